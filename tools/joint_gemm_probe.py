"""Diagnostic: do two concurrent streams recover the tile-round quantisation of the N = 768 token GEMMs, or would ONE launch over the rows of both
student passes (global 50 432 + local 51 200 rows) be faster?  Times per shape: the two launches one after the other on one stream, the two on
two streams, one launch over all rows."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lightly_train_amd  # noqa
from lightly_train_amd import ops
dev = "cuda"
D = 768
Tg, Tl = 256 * 197, 1024 * 50
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

def mk(M, N, K, epi, tb):
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    B = torch.randn((K, N) if tb else (N, K), device=dev).to(torch.bfloat16)
    f32 = epi == ops.EPI_RESID
    C = torch.zeros(M, N, device=dev, dtype=torch.float32 if f32 else torch.bfloat16)
    kw = {}
    if epi == ops.EPI_RESID:
        kw = dict(bias=torch.zeros(N, device=dev), gamma=torch.ones(N, device=dev), resid=torch.zeros(M, N, device=dev))
    if epi == ops.EPI_BF16_GELU:
        kw = dict(bias=torch.zeros(N, device=dev), out2=torch.empty(M, N, device=dev, dtype=torch.bfloat16))
    if epi == ops.EPI_BF16_GELUGRAD:
        kw = dict(aux=torch.zeros(M, N, device=dev, dtype=torch.bfloat16))
    return lambda: ops.gemm(A, B, C, M=M, N=N, K=K, trans_b=tb, epilogue=epi, **kw)

def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

for name, N, K, epi, tb in (("proj fwd resid", D, D, ops.EPI_RESID, False), ("fc2 fwd resid", D, 4 * D, ops.EPI_RESID, False), ("qkv dgrad", D, 3 * D, ops.EPI_BF16, True),
                            ("fc1 dgrad", D, 4 * D, ops.EPI_BF16, True), ("qkv fwd", 3 * D, D, ops.EPI_BF16, False), ("fc1 fwd gelu", 4 * D, D, ops.EPI_BF16_GELU, False),
                            ("fc2 dgrad gelugrad", 4 * D, D, ops.EPI_BF16_GELUGRAD, True)):
    g, l, j = mk(Tg, N, K, epi, tb), mk(Tl, N, K, epi, tb), mk(Tg + Tl, N, K, epi, tb)
    def serial(): g(); l()
    def conc():
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur); s2.wait_stream(cur)
        with torch.cuda.stream(s1): g()
        with torch.cuda.stream(s2): l()
        cur.wait_stream(s1); cur.wait_stream(s2)
    a, b, c = timeit(serial), timeit(conc), timeit(j)
    fl = 2.0 * (Tg + Tl) * N * K
    print(f"{name:20s} N={N:5d} K={K:5d}: serial {a:7.1f} us ({fl / a / 1e6:6.0f} TF/s) | two streams {b:7.1f} us ({fl / b / 1e6:6.0f}) | one joint launch {c:7.1f} us ({fl / c / 1e6:6.0f})")
