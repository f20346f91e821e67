"""Persistent GEMM (force_kernel 10, gemm_p.hip) against the four-phase kernel (force_kernel 8) and a torch fp32 reference on the step's
token shapes: correctness of every epilogue it serves (bf16-rounded branch output, rows past M, column tiles past N) and back-to-back
timing of both kernels.  usage: python tools/gemm1p_check.py [check|bench|all] [M]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lightly_train_amd  # noqa: F401
from lightly_train_amd import ops

dev = "cuda"


def make(M, N, K, tb, epi, seed=0):
    g = torch.Generator(device=dev).manual_seed(seed)
    A = (torch.randn(M, K, device=dev, generator=g) * 0.5).to(torch.bfloat16)
    B = (torch.randn((K, N) if tb else (N, K), device=dev, generator=g) * 0.05).to(torch.bfloat16)
    kw = {}
    if epi in (ops.EPI_BF16, ops.EPI_BF16_GELU, ops.EPI_RESID):
        kw["bias"] = torch.randn(N, device=dev, generator=g) * 0.1
    if epi == ops.EPI_BF16_GELU:
        kw["out2"] = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    if epi == ops.EPI_RESID:
        kw["gamma"] = torch.randn(N, device=dev, generator=g)
        kw["resid"] = torch.randn(M, N, device=dev, generator=g)
        kw["rowscale"] = torch.rand(M, device=dev, generator=g) + 0.5
        kw["branch_scale"] = 1.25
    if epi == ops.EPI_BF16_GELUGRAD:
        kw["aux"] = torch.randn(M, N, device=dev, generator=g).to(torch.bfloat16)
    return A, B, kw


def run(A, B, kw, M, N, K, tb, epi, fk, alpha=1.0):
    f32 = epi == ops.EPI_RESID
    C = torch.full((M, N), float("nan"), device=dev, dtype=torch.float32 if f32 else torch.bfloat16)
    kw = dict(kw)
    if "out2" in kw:
        kw["out2"] = torch.full_like(kw["out2"], float("nan"))
    ops.gemm(A, B, C, M=M, N=N, K=K, trans_b=tb, epilogue=epi, force_kernel=fk, alpha=alpha, **kw)
    return C, kw.get("out2")


def reference(A, B, kw, tb, epi, alpha=1.0):
    acc = A.float() @ (B.float() if tb else B.float().t())
    v = (acc * alpha + (kw["bias"] if "bias" in kw else 0.0)).to(torch.bfloat16).float()   # the Linear's bf16 output (autocast)
    if epi == ops.EPI_BF16:
        return v, None
    if epi == ops.EPI_BF16_GELU:
        return torch.nn.functional.gelu(v), v
    if epi == ops.EPI_RESID:
        return kw["resid"] + kw["branch_scale"] * kw["rowscale"][:, None] * kw["gamma"][None, :] * v, None
    if epi == ops.EPI_BF16_GELUGRAD:
        x = kw["aux"].float()
        gp = 0.5 * (1 + torch.erf(x / 2 ** 0.5)) + x * torch.exp(-0.5 * x * x) / (2 * torch.pi) ** 0.5
        return v * gp, None


def check():
    bad = 0
    shapes = [(2048, 768, 768), (2000, 768, 256), (4099, 2304, 768), (1984, 3072, 768), (2048, 768, 3072), (2048, 320, 384), (6304, 384, 1536),
              (50432, 768, 768)]
    for (M, N, K) in shapes:
        for tb in (False, True):
            for epi in (ops.EPI_BF16, ops.EPI_BF16_GELU, ops.EPI_RESID, ops.EPI_BF16_GELUGRAD):
                if M > 10000 and epi != ops.EPI_RESID:
                    continue
                A, B, kw = make(M, N, K, tb, epi, seed=M + N + K)
                c10, c2_10 = run(A, B, kw, M, N, K, tb, epi, 10)
                torch.cuda.synchronize()
                ref, ref2 = reference(A, B, kw, tb, epi)
                c8, _ = run(A, B, kw, M, N, K, tb, epi, 8 if K % 64 == 0 and N >= 256 else 1)
                def err(x, r):
                    x = x.float()
                    if not torch.isfinite(x).all():
                        return float("inf")
                    return ((x - r).abs().max() / r.abs().max()).item()
                e10, e8 = err(c10, ref), err(c8, ref)
                e2 = err(c2_10, ref2) if ref2 is not None else 0.0
                ok = e10 < 1.2e-2 and e2 < 1.2e-2
                bad += not ok
                if not ok:
                    d = ((c10.float() - ref).abs() > 0.02 * ref.abs().max())
                    rows = d.any(1).nonzero().flatten(); cols = d.any(0).nonzero().flatten()
                    print("   wrong elements", int(d.sum()), "of", d.numel(), "rows", rows[:6].tolist(), "..", rows[-3:].tolist(), "n_rows", rows.numel(),
                          "cols", cols[:4].tolist(), "..", cols[-2:].tolist(), "n_cols", cols.numel(),
                          "row blocks(192)", sorted(set((rows // 192).tolist()))[:12], "nan", int((~torch.isfinite(c10.float())).sum()))
                print(f"{'ok ' if ok else 'BAD'} M={M:6d} N={N:5d} K={K:5d} tb={int(tb)} epi={epi}: 1p {e10:.2e} (C2 {e2:.1e})  q {e8:.2e}", flush=True)
    print("FAILURES:", bad)
    return bad


def bench(T=256 * 197, iters=20):
    D = 768
    cases = [("qkv fwd", T, 3 * D, D, False, ops.EPI_BF16), ("proj fwd resid", T, D, D, False, ops.EPI_RESID),
             ("fc1 fwd gelu", T, 4 * D, D, False, ops.EPI_BF16_GELU), ("fc2 fwd resid", T, D, 4 * D, False, ops.EPI_RESID),
             ("fc2 dgrad gelugrad", T, 4 * D, D, True, ops.EPI_BF16_GELUGRAD), ("fc1 dgrad", T, D, 4 * D, True, ops.EPI_BF16),
             ("qkv dgrad", T, D, 3 * D, True, ops.EPI_BF16), ("proj dgrad", T, D, D, True, ops.EPI_BF16)]
    for name, M, N, K, tb, epi in cases:
        A, B, kw = make(M, N, K, tb, epi)
        f32 = epi == ops.EPI_RESID
        C = torch.zeros(M, N, device=dev, dtype=torch.float32 if f32 else torch.bfloat16)
        res = {}
        for rnd in range(3):
            for fk in (8, 10):
                for _ in range(2):
                    ops.gemm(A, B, C, M=M, N=N, K=K, trans_b=tb, epilogue=epi, force_kernel=fk, **kw)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(iters):
                    ops.gemm(A, B, C, M=M, N=N, K=K, trans_b=tb, epilogue=epi, force_kernel=fk, **kw)
                e1.record(); torch.cuda.synchronize()
                res.setdefault(fk, []).append(e0.elapsed_time(e1) / iters * 1e3)
        q, p = min(res[8]), min(res[10])
        print(f"{name:22s} M={M} N={N:5d} K={K:5d}: q {q:7.1f} us ({2*M*N*K/q/1e6:6.1f} TF/s)   1p {p:7.1f} us ({2*M*N*K/p/1e6:6.1f} TF/s)   {q/p:5.2f}x", flush=True)


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "all"
    rc = 0
    if mode in ("check", "all"):
        rc = check()
    if mode in ("bench", "all"):
        bench(int(sys.argv[2]) if len(sys.argv) > 2 else 256 * 197)
    sys.exit(1 if rc else 0)
