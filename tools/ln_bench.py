"""LayerNorm forward / backward kernels in isolation: time and HBM rate at the step's shapes.  usage: python tools/ln_bench.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lightly_train_amd  # noqa: E402,F401
from lightly_train_amd import ops  # noqa: E402


def timeit(fn, iters=30, flush=None):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        if flush is not None:
            flush.add_(1.0)     # 1 GiB pass: evict the operands from the Infinity Cache, as a step's intervening GEMMs do
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def main():
    dev = "cuda"
    flush = torch.zeros(256 * 1024 * 1024, device=dev)
    for T, D in ((25216, 768), (18944, 768), (25216, 1024), (6304, 384)):
        x = torch.randn(T, D, device=dev)
        w, b = torch.randn(D, device=dev), torch.randn(D, device=dev)
        y = torch.empty(T, D, device=dev, dtype=torch.bfloat16)
        mean, rstd = torch.empty(T, device=dev), torch.empty(T, device=dev)
        f = lambda: ops.layernorm_fwd(x, w, b, T, D, y_bf16=y, mean=mean, rstd=rstd)
        for R in ("0", "1", "2", "4"):
            os.environ["LT_LN_FWD_ROWS"] = R
            for name, fl in (("warm", None), ("cold", flush)):
                t = timeit(f, flush=fl)
                print(f"fwd R={R} T={T} D={D} {name}: {t:7.1f} us  {(T * D * 6 + T * 8) / t / 1e6:6.2f} TB/s")
        os.environ.pop("LT_LN_FWD_ROWS")
        for name, fl in (("warm", None), ("cold", flush)):   # the same traffic without the row reductions: torch's cast kernel
            t = timeit(lambda: y.copy_(x), flush=fl)
            print(f"cast(torch)   T={T} D={D} {name}: {t:7.1f} us  {(T * D * 6) / t / 1e6:6.2f} TB/s")
        dy = torch.randn(T, D, device=dev).bfloat16()
        dres = torch.randn(T, D, device=dev)
        dx = torch.empty(T, D, device=dev)
        dw, db = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
        dnext = torch.empty(T, D, device=dev, dtype=torch.bfloat16)
        gam, dbn = torch.randn(D, device=dev), torch.zeros(D, device=dev)
        g1 = lambda: ops.layernorm_bwd(x, w, mean, rstd, dy, dres, dx, dw, db, T, D)
        g2 = lambda: ops.layernorm_bwd(x, w, mean, rstd, dy, dres, dx, dw, db, T, D, dnext=dnext, gamma_next=gam, dbias_next=dbn)
        for nm, fn, byts in (("bwd ", g1, T * D * 14), ("bwd+", g2, T * D * 16)):
            for name, fl in (("warm", None), ("cold", flush)):
                t = timeit(fn, flush=fl)
                print(f"{nm} T={T} D={D} {name}: {t:7.1f} us  {byts / t / 1e6:6.2f} TB/s")


if __name__ == "__main__":
    main()
