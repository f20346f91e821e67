"""Data-gradient GEMMs dX = dY W: W as stored ([out][in] = [K][N], the transposing-fragment-read form, trans_b) against a transposed copy W^T ([N][K], the
forward form) on the step's shapes -- would a per-step transposed bf16 weight copy pay?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lightly_train_amd  # noqa
from lightly_train_amd import ops
dev = "cuda"
def t_of(f, iters=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for M in (50432, 37888):
    for name, N, K, epi in (("proj dgrad", 768, 768, ops.EPI_BF16), ("qkv dgrad", 768, 2304, ops.EPI_BF16), ("fc1 dgrad", 768, 3072, ops.EPI_BF16), ("fc2 dgrad + GELU'", 3072, 768, ops.EPI_BF16_GELUGRAD)):
        A = torch.randn(M, K, device=dev).to(torch.bfloat16)
        W = (torch.randn(K, N, device=dev) * 0.02).to(torch.bfloat16)
        Wt = W.t().contiguous()
        C1 = torch.empty(M, N, device=dev, dtype=torch.bfloat16); C2 = torch.empty_like(C1)
        aux = torch.randn(M, N, device=dev).to(torch.bfloat16) if epi == ops.EPI_BF16_GELUGRAD else None
        kw = dict(aux=aux) if aux is not None else {}
        try:
            f1 = lambda: ops.gemm(A, W, C1, M=M, N=N, K=K, trans_b=True, epilogue=epi, **kw)
            f2 = lambda: ops.gemm(A, Wt, C2, M=M, N=N, K=K, trans_b=False, epilogue=epi, **kw)
            t1, t2 = t_of(f1), t_of(f2)
            same = torch.equal(C1, C2)
            print(f"M {M} {name:18s} N {N:5d} K {K:5d}: W as stored {t1:7.1f} us | W^T copy {t2:7.1f} us | {100 * (t2 / t1 - 1):+5.1f} % | bit-identical {same}")
        except Exception as e:  # noqa
            print(name, "failed:", repr(e)[:200])
