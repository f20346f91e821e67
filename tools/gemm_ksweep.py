"""Fixed cost per tile round of the token GEMM kernel: T(K) = a + b K at a shape of exactly `rounds` x 256 tiles (one tile per CU and
round), per epilogue.  a / rounds = workgroup launch + pipeline fill + epilogue, none of which overlaps another tile's MFMA work (one
workgroup per CU)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lightly_train_amd  # noqa
from lightly_train_amd import ops

def t_us(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

N = 1024
for rounds in (1, 2, 4):
    M = rounds * 64 * 256
    for epi, name in ((ops.EPI_BF16, "bf16"), (ops.EPI_BF16_GELU, "gelu+pre"), (ops.EPI_RESID, "resid")):
        res = []
        for K in (256, 512, 1024, 2048, 4096):
            a = torch.randn(M, K, device="cuda").to(torch.bfloat16); b = torch.randn(N, K, device="cuda").to(torch.bfloat16)
            f32 = epi == ops.EPI_RESID
            out = torch.empty(M, N, device="cuda", dtype=torch.float32 if f32 else torch.bfloat16)
            kw = {}
            if epi == ops.EPI_BF16_GELU: kw = dict(bias=torch.zeros(N, device="cuda"), out2=torch.empty_like(out))
            if epi == ops.EPI_RESID: kw = dict(bias=torch.zeros(N, device="cuda"), gamma=torch.ones(N, device="cuda"), resid=torch.randn(M, N, device="cuda"))
            res.append((K, t_us(lambda: ops.gemm(a, b, out, M=M, N=N, K=K, epilogue=epi, **kw))))
        (k1, t1), (k2, t2) = res[-2], res[-1]
        slope = (t2 - t1) / (k2 - k1)
        icpt = t2 - slope * k2
        per_ktile_us = slope * 64 / rounds                                      # one 256 x 256 x 64 K-tile on every CU
        peak = 256 * 2.0 * 256 * 256 * 64 / per_ktile_us / 1e6 if slope > 0 else 0   # TF/s of the steady-state K loop, whole chip
        print(f"rounds={rounds} {name:9s}: " + "  ".join(f"K={k}:{t:7.1f}" for k, t in res) + f"   us/64-K per round {slope * 64 / rounds:.3f}  fixed/round {icpt / rounds:6.2f} us  loop {peak:6.0f} TF/s")
