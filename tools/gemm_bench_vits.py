"""GEMM micro-benchmark over the ViT-S/16 step's shapes (D = 384, per-GPU batch 128: 50 432 global / 51 200 local token rows),
each shape with a per-call dispatch switch alternating launch by launch in one process (medians).

The two switches it flips (LT_GEMM_NSPLIT: 256-wide kernel on the first 256 m columns + 128-wide kernel on the rest;
LT_GEMM_SCORE_COLS: wgrad tile scoring that counts the empty columns of the last tile) existed in a throw-away build of
`lt_gemm_bf16`'s dispatcher only -- the result (profiles/r02m_gemm_vits_dispatch_ab.log: mixed, not shipped) is why; with the shipped
library both columns time the same kernel and the tool is a plain ViT-S shape benchmark.

  python tools/gemm_bench_vits.py
"""
import os, statistics, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lightly_train_amd  # noqa: F401
from lightly_train_amd import ops

WS = torch.empty(64 * 1024 * 1024, device="cuda")


def bench(name, M, N, K, ta, tb, epi, env, split=1, ws=None):
    dev = "cuda"
    A = torch.randn((K, M) if ta else (M, K), device=dev).to(torch.bfloat16)
    B = torch.randn((K, N) if tb else (N, K), device=dev).to(torch.bfloat16)
    f32 = epi in (ops.EPI_RESID, ops.EPI_F32, ops.EPI_F32_ACCUM)
    C = torch.zeros(M, N, device=dev, dtype=torch.float32 if f32 else torch.bfloat16)
    kw = {}
    if epi == ops.EPI_BF16_GELU:
        kw = dict(bias=torch.zeros(N, device=dev), out2=torch.empty(M, N, device=dev, dtype=torch.bfloat16))
    if epi == ops.EPI_RESID:
        kw = dict(bias=torch.zeros(N, device=dev), gamma=torch.ones(N, device=dev), resid=torch.zeros(M, N, device=dev))
    if epi == ops.EPI_BF16_GELUGRAD:
        kw = dict(aux=torch.zeros(M, N, device=dev, dtype=torch.bfloat16))
    t = {0: [], 1: []}
    for i in range(24):
        v = i & 1
        os.environ[env] = str(v)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.gemm(A, B, C, M=M, N=N, K=K, trans_a=ta, trans_b=tb, epilogue=epi, split_k=split, workspace=ws, **kw)
        e1.record(); torch.cuda.synchronize()
        if i >= 4:
            t[v].append(e0.elapsed_time(e1) * 1e3)
    m0, m1 = statistics.median(t[0]), statistics.median(t[1])
    print(f"{name:22s} M={M:6d} N={N:5d} K={K:6d}  {env}=0: {m0:7.1f} us ({2*M*N*K/m0/1e6:6.1f} TF/s)   =1: {m1:7.1f} us ({2*M*N*K/m1/1e6:6.1f} TF/s)")


T, D = 256 * 197, 384
E = "LT_GEMM_NSPLIT"
bench("qkv fwd", T, 3 * D, D, False, False, ops.EPI_BF16, E)
bench("proj fwd resid", T, D, D, False, False, ops.EPI_RESID, E)
bench("fc1 fwd gelu", T, 4 * D, D, False, False, ops.EPI_BF16_GELU, E)
bench("fc2 fwd resid", T, D, 4 * D, False, False, ops.EPI_RESID, E)
bench("fc2 dgrad gelugrad", T, 4 * D, D, False, True, ops.EPI_BF16_GELUGRAD, E)
bench("fc1 dgrad", T, D, 4 * D, False, True, ops.EPI_BF16, E)
bench("qkv dgrad", T, D, 3 * D, False, True, ops.EPI_BF16, E)
bench("proj dgrad", T, D, D, False, True, ops.EPI_BF16, E)
E = "LT_GEMM_SCORE_COLS"
for nm, mm, nn in (("fc1 wgrad", 4 * D, D), ("fc2 wgrad", D, 4 * D), ("qkv wgrad", 3 * D, D), ("proj wgrad", D, D)):
    bench(nm + " slab", mm, nn, T, True, True, ops.EPI_F32_ACCUM, E, split=0, ws=WS)
