"""Time the hand-written GEMM (lt_gemm_bf16, four-phase 256x256 kernel) against the ROCm library GEMM behind torch.matmul (hipBLASLt /
rocBLAS) on the step's shapes -- a yardstick for the kernel, not a product path.  usage: python tools/gemm_vs_library.py [vitb|vitl]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lightly_train_amd  # noqa: E402,F401
from lightly_train_amd import ops  # noqa: E402


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    D, H = (1024, 4096) if (len(sys.argv) > 1 and sys.argv[1] == "vitl") else (768, 3072)
    T = 25216
    dev = "cuda"
    g = torch.Generator().manual_seed(0)
    rows = []
    for name, M, N, K, tb in [("qkv fwd", T, 3 * D, D, False), ("proj fwd", T, D, D, False), ("fc1 fwd", T, H, D, False), ("fc2 fwd", T, D, H, False),
                              ("qkv dgrad", T, D, 3 * D, True), ("fc1 dgrad", T, D, H, True), ("fc2 dgrad", T, H, D, True)]:
        a = torch.randn(M, K, generator=g).to(dev).bfloat16()
        # forward: W [N, K] (C = A W^T); dgrad: W [K, N] (C = A W), trans_b
        w = (torch.randn(K, N, generator=g) if tb else torch.randn(N, K, generator=g)).to(dev).bfloat16()
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        t_ours = timeit(lambda: ops.gemm(a, w, out, M=M, N=N, K=K, trans_b=tb, epilogue=ops.EPI_BF16))
        ref = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        wt = w if tb else w.t()
        t_lib = timeit(lambda: torch.matmul(a, wt, out=ref))
        err = ((out.float() - ref.float()).abs().max() / ref.float().abs().max()).item()
        fl = 2.0 * M * N * K
        rows.append((name, M, N, K, t_ours, fl / t_ours / 1e6, t_lib, fl / t_lib / 1e6, err))
    # weight gradient: dW [N_out, K_in] = dY^T X over T rows
    for name, n_out, k_in in [("qkv wgrad", 3 * D, D), ("fc1 wgrad", H, D), ("fc2 wgrad", D, H)]:
        dy = (torch.randn(T, n_out, generator=g) * 0.01).to(dev).bfloat16()
        x = torch.randn(T, k_in, generator=g).to(dev).bfloat16()
        out = torch.zeros(n_out, k_in, device=dev)
        slab = torch.empty(32 * 1024 * 1024, device=dev)
        t_ours = timeit(lambda: ops.gemm(dy, x, out, M=n_out, N=k_in, K=T, trans_a=True, trans_b=True, epilogue=ops.EPI_F32_ACCUM, split_k=2, lda=n_out, ldb=k_in,
                                         ldc=k_in, workspace=slab))
        ref = torch.empty(n_out, k_in, device=dev, dtype=torch.bfloat16)
        t_lib = timeit(lambda: torch.matmul(dy.t(), x, out=ref))
        fl = 2.0 * T * n_out * k_in
        rows.append((name, n_out, k_in, T, t_ours, fl / t_ours / 1e6, t_lib, fl / t_lib / 1e6, float("nan")))
    print("| GEMM | M | N | K | ours us | ours TF/s | library us | library TF/s | max rel diff |\n|---|---|---|---|---|---|---|---|---|")
    for r in rows:
        print(f"| {r[0]} | {r[1]} | {r[2]} | {r[3]} | {r[4]:.1f} | {r[5]:.0f} | {r[6]:.1f} | {r[7]:.0f} | {r[8]:.1e} |")


if __name__ == "__main__":
    main()
