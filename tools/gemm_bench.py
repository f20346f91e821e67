"""GEMM micro-benchmark over the shapes of the ViT-B/16 step (per-GPU batch 128)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lightly_train_amd
from lightly_train_amd import ops

def bench(name, M, N, K, ta, tb, epi, split=1, iters=10, fk=0, ws=None):
    dev = "cuda"
    A = torch.randn((K, M) if ta else (M, K), device=dev).to(torch.bfloat16)
    B = torch.randn((K, N) if tb else (N, K), device=dev).to(torch.bfloat16)
    f32 = epi in (ops.EPI_RESID, ops.EPI_F32, ops.EPI_F32_ACCUM)
    C = torch.zeros(M, N, device=dev, dtype=torch.float32 if f32 else torch.bfloat16)
    kw = {}
    if epi == ops.EPI_BF16_GELU:
        kw = dict(bias=torch.zeros(N, device=dev), out2=torch.empty(M, N, device=dev, dtype=torch.bfloat16))
    if epi == ops.EPI_RESID:
        kw = dict(bias=torch.zeros(N, device=dev), gamma=torch.ones(N, device=dev), resid=torch.zeros(M, N, device=dev),
                  out2=torch.empty(M, N, device=dev, dtype=torch.bfloat16))
    if epi == ops.EPI_BF16_GELUGRAD:
        kw = dict(aux=torch.zeros(M, N, device=dev, dtype=torch.bfloat16))
    for _ in range(2):
        ops.gemm(A, B, C, M=M, N=N, K=K, trans_a=ta, trans_b=tb, epilogue=epi, split_k=split, force_kernel=fk, workspace=ws, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.gemm(A, B, C, M=M, N=N, K=K, trans_a=ta, trans_b=tb, epilogue=epi, split_k=split, force_kernel=fk, workspace=ws, **kw)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"{name:30s} k{fk} M={M:6d} N={N:6d} K={K:6d} ta={int(ta)} tb={int(tb)} epi={epi} split={split:2d}: {ms*1e3:8.1f} us  {2*M*N*K/ms/1e9:7.1f} TF/s")

T = 256 * 197
D = 768
FKS = tuple(int(x) for x in sys.argv[1].split(',')) if len(sys.argv) > 1 else (2, 5)
for fk in (FKS if not (len(sys.argv) > 2 and sys.argv[2] in ('tok-only', 'all')) else ()):
    bench("square 4096 NN bf16", 4096, 4096, 4096, False, False, ops.EPI_BF16, fk=fk)
    bench("square 8192 NN bf16", 8192, 8192, 8192, False, False, ops.EPI_BF16, fk=fk)
    bench("square 4096 NT(tb) bf16", 4096, 4096, 4096, False, True, ops.EPI_BF16, fk=fk)
    bench("qkv fwd", T, 3 * D, D, False, False, ops.EPI_BF16, fk=fk)
    bench("proj fwd resid", T, D, D, False, False, ops.EPI_RESID, fk=fk)
    bench("fc1 fwd gelu", T, 4 * D, D, False, False, ops.EPI_BF16_GELU, fk=fk)
    bench("fc2 fwd resid", T, D, 4 * D, False, False, ops.EPI_RESID, fk=fk)
    bench("fc2 dgrad gelugrad", T, 4 * D, D, False, True, ops.EPI_BF16_GELUGRAD, fk=fk)
    bench("fc1 dgrad", T, D, 4 * D, False, True, ops.EPI_BF16, fk=fk)
    bench("qkv dgrad", T, D, 3 * D, False, True, ops.EPI_BF16, fk=fk)
    bench("head last fwd", 8832, 65536, 256, False, False, ops.EPI_F32, fk=fk)
    bench("head last dgrad", 8832, 256, 65536, False, True, ops.EPI_F32, fk=fk)
if len(sys.argv) > 2 and sys.argv[2] == "tok-only":
    for fk in FKS:
        bench("qkv fwd", T, 3 * D, D, False, False, ops.EPI_BF16, fk=fk)
        bench("proj fwd resid", T, D, D, False, False, ops.EPI_RESID, fk=fk)
        bench("fc1 fwd gelu", T, 4 * D, D, False, False, ops.EPI_BF16_GELU, fk=fk)
        bench("fc2 fwd resid", T, D, 4 * D, False, False, ops.EPI_RESID, fk=fk)
        bench("fc2 dgrad gelugrad", T, 4 * D, D, False, True, ops.EPI_BF16_GELUGRAD, fk=fk)
        bench("fc1 dgrad", T, D, 4 * D, False, True, ops.EPI_BF16, fk=fk)
    sys.exit(0)
if len(sys.argv) > 2 and sys.argv[2] == "all":
    WS = torch.empty(64 * 1024 * 1024, device="cuda")
    for fk in FKS:
        bench("square 4096 NN bf16", 4096, 4096, 4096, False, False, ops.EPI_BF16, fk=fk)
        bench("square 8192 NN bf16", 8192, 8192, 8192, False, False, ops.EPI_BF16, fk=fk)
        bench("square 4096 NT(tb) bf16", 4096, 4096, 4096, False, True, ops.EPI_BF16, fk=fk)
        bench("square 4096 TT f32acc", 4096, 4096, 4096, True, True, ops.EPI_F32_ACCUM, fk=fk)
        bench("qkv fwd", T, 3 * D, D, False, False, ops.EPI_BF16, fk=fk)
        bench("proj fwd resid", T, D, D, False, False, ops.EPI_RESID, fk=fk)
        bench("fc1 fwd gelu", T, 4 * D, D, False, False, ops.EPI_BF16_GELU, fk=fk)
        bench("fc2 fwd resid", T, D, 4 * D, False, False, ops.EPI_RESID, fk=fk)
        bench("fc2 dgrad gelugrad", T, 4 * D, D, False, True, ops.EPI_BF16_GELUGRAD, fk=fk)
        bench("fc1 dgrad", T, D, 4 * D, False, True, ops.EPI_BF16, fk=fk)
        bench("qkv dgrad", T, D, 3 * D, False, True, ops.EPI_BF16, fk=fk)
        for nm, mm, nn in (("fc1 wgrad", 4 * D, D), ("fc2 wgrad", D, 4 * D), ("qkv wgrad", 3 * D, D), ("proj wgrad", D, D)):
            bench(nm + " slab", mm, nn, T, True, True, ops.EPI_F32_ACCUM, split=0, fk=fk, ws=WS)
    sys.exit(0)
if len(sys.argv) > 2 and sys.argv[2] == "fwd-only":
    sys.exit(0)
bench("square 4096 TT f32acc", 4096, 4096, 4096, True, True, ops.EPI_F32_ACCUM)
for s in (2, 8):
    bench("fc1 wgrad", 4 * D, D, T, True, True, ops.EPI_F32_ACCUM, split=s, fk=1)
bench("proj wgrad", D, D, T, True, True, ops.EPI_F32_ACCUM, split=8, fk=1)
WS = torch.empty(64 * 1024 * 1024, device="cuda")
for nm, mm, nn in (("fc1 wgrad", 4 * D, D), ("fc2 wgrad", D, 4 * D), ("qkv wgrad", 3 * D, D), ("proj wgrad", D, D)):
    bench(nm + " slab", mm, nn, T, True, True, ops.EPI_F32_ACCUM, split=2, fk=2, ws=WS)
    bench(nm + " atomic", mm, nn, T, True, True, ops.EPI_F32_ACCUM, split=2, fk=2)
bench("head last wgrad", 65536, 256, 8832, True, True, ops.EPI_F32_ACCUM, split=1)
