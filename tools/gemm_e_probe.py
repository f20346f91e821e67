"""The static-address K-loop kernel (force_kernel = 11, gemm256e) against the four-phase kernel it derives from (force_kernel = 8, gemm256q) on the
step's GEMM shapes with their real epilogues: bitwise equality of every output (same MFMA order => same bits) and interleaved timing rounds
in one process (median of rounds, us per launch and TF/s), plus the loop-dominated 4096 x 4096 x 8192 shape and the ROCm library's rate.
usage: python tools/gemm_e_probe.py [--rounds 5] [--iters 10] [--lib 1]"""
import argparse
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lightly_train_amd  # noqa: E402,F401
from lightly_train_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--lib", type=int, default=1)
ap.add_argument("--kernels", default="8,11t0,11")
ap.add_argument("--quick", type=int, default=0)
a = ap.parse_args()
KERNELS = a.kernels.split(",")     # "8", "11" (two phases per K-tile, the default), "11p4" (= force_kernel 11 with LT_GEMM_E_PH=4)
dev = "cuda"
g = torch.Generator().manual_seed(0)


def rnd(*shape, scale=1.0, dtype=torch.bfloat16):
    return (torch.randn(*shape, generator=g) * scale).to(dev).to(dtype)


def call(spec, f):
    """spec "11p4": force_kernel 11 under LT_GEMM_E_PH=4; "11t0": without the opt-in 128-row tail kernel (LT_GEMM_TAIL128=0; every other spec runs
    WITH it).  The library reads the variables per call."""
    import re
    m = re.match(r"^(\d+)(?:p(\d))?(t0)?$", spec)
    fk, ph, t0 = m.group(1), m.group(2), m.group(3)
    for var, val in (("LT_GEMM_E_PH", ph), ("LT_GEMM_TAIL128", "0" if t0 else "1")):
        if val:
            os.environ[var] = val
        else:
            os.environ.pop(var, None)
    return f(int(fk))


def time_rounds(fns):
    """fns: {name: callable}; interleaved rounds; returns {name: median us}."""
    for f in fns.values():
        for _ in range(2):
            f()
    torch.cuda.synchronize()
    res = {k: [] for k in fns}
    for _ in range(a.rounds):
        for k, f in fns.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                f()
            e1.record()
            torch.cuda.synchronize()
            res[k].append(e0.elapsed_time(e1) / a.iters * 1e3)
    return {k: statistics.median(v) for k, v in res.items()}


def case(name, M, N, K, *, tb=False, epi=ops.EPI_BF16, wgrad=False):
    fl = 2.0 * M * N * K
    outs, fns = {}, {}
    if wgrad:   # dW [M = n_out, N = k_in] += dY^T X over K = tokens; slab split-K + column sums through the ledger-less path (colsum separate)
        dy, x = rnd(K, M, scale=0.01), rnd(K, N)
        slab = torch.empty(48 * 1024 * 1024, device=dev)
        for fk in KERNELS:
            out = torch.zeros(M, N, device=dev)
            outs[fk] = out
            fns[fk] = (lambda fk=fk, out=out: call(fk, lambda k: ops.gemm(dy, x, out, M=M, N=N, K=K, trans_a=True, trans_b=True, epilogue=ops.EPI_F32_ACCUM, split_k=0, lda=M, ldb=N,
                                                                           ldc=N, workspace=slab, force_kernel=k)))
        lib = (lambda: torch.matmul(dy.t(), x)) if a.lib else None
    else:
        A = rnd(M, K)
        W = rnd(K, N) if tb else rnd(N, K)
        bias = rnd(N, dtype=torch.float32) if epi != ops.EPI_BF16_GELUGRAD else None
        gamma = rnd(N, dtype=torch.float32) if epi == ops.EPI_RESID else None
        resid = rnd(M, N, dtype=torch.float32) if epi == ops.EPI_RESID else None
        aux = rnd(M, N) if epi == ops.EPI_BF16_GELUGRAD else None
        f32 = epi in (ops.EPI_RESID, ops.EPI_F32)
        for fk in KERNELS:
            out = torch.empty(M, N, device=dev, dtype=torch.float32 if f32 else torch.bfloat16)
            out2 = torch.empty(M, N, device=dev, dtype=torch.bfloat16) if epi == ops.EPI_BF16_GELU else None
            outs[fk] = (out, out2)
            fns[fk] = (lambda fk=fk, out=out, out2=out2: call(fk, lambda k: ops.gemm(A, W, out, M=M, N=N, K=K, trans_b=tb, epilogue=epi, bias=bias, gamma=gamma, resid=resid, aux=aux,
                                                                                      out2=out2, force_kernel=k)))
        Wt = W if tb else W.t()
        lib = (lambda: torch.matmul(A, Wt)) if a.lib else None
    # equality (wgrad: run once from zero)
    for fk in KERNELS:
        if wgrad:
            outs[fk].zero_()
        fns[fk]()
    torch.cuda.synchronize()
    ref = outs[KERNELS[0]]
    same = []
    for fk in KERNELS[1:]:
        o = outs[fk]
        if wgrad:
            same.append(bool(torch.equal(o, ref)))
        else:
            same.append(bool(torch.equal(o[0], ref[0]) and (o[1] is None or torch.equal(o[1], ref[1]))))
    allf = dict(fns)
    if lib is not None:
        allf["lib"] = lib
    t = time_rounds(allf)
    cols = "  ".join(f"k{k}: {t[k]:7.1f} us {fl / t[k] / 1e6:5.0f} TF" for k in KERNELS)
    libs = f"  lib: {t['lib']:7.1f} us {fl / t['lib'] / 1e6:6.0f} TF" if lib is not None else ""
    print(f"{name:28s} M {M:6d} N {N:5d} K {K:6d}  {cols}{libs}  bit-equal {same}", flush=True)


T, D, H = 50432, 768, 3072
case("loop NN 4096x4096x8192", 4096, 4096, 8192)
case("loop NT 4096x4096x8192", 4096, 4096, 8192, tb=True)
if not a.quick:
    case("qkv fwd (bias)", T, 3 * D, D)
    case("proj fwd (resid)", T, D, D, epi=ops.EPI_RESID)
    case("fc1 fwd (gelu + pre)", T, H, D, epi=ops.EPI_BF16_GELU)
    case("fc2 fwd (resid)", T, D, H, epi=ops.EPI_RESID)
    case("fc2 dgrad (gelu')", T, H, D, tb=True, epi=ops.EPI_BF16_GELUGRAD)
    case("fc1 dgrad", T, D, H, tb=True)
    case("proj dgrad", T, D, D, tb=True)
    case("qkv dgrad", T, D, 3 * D, tb=True)
    case("proj fwd 25216 (plain)", 25216, D, D)
    case("fc1 fwd 25216 (plain)", 25216, H, D)
    case("fc2 fwd 25216 (plain)", 25216, D, H)
    case("qkv wgrad", 3 * D, D, T, wgrad=True)
    case("fc1 wgrad", H, D, T, wgrad=True)
    case("fc2 wgrad", D, H, T, wgrad=True)
    case("logits (f32)", 4096, 65536, 256, epi=ops.EPI_F32)
    case("vit-l fc2 fwd", 25216, 1024, 4096)
    case("vit-s qkv fwd", 50432, 1152, 384)
