"""Race screen for the persistent GEMM (gemm_p.hip): the shapes whose tails exercise clamped / redirected DMA pieces, every epilogue, repeated
with different data while another stream keeps the memory system busy.  usage: python tools/gemm1p_stress.py [rounds]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import lightly_train_amd  # noqa: F401
from lightly_train_amd import ops
import gemm1p_check as G

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
only = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else None
shapes = [(4099, 2304, 768), (2000, 768, 256), (2048, 320, 384), (6304, 384, 1536), (9650, 768, 3072)]
side = torch.cuda.Stream()
junk = torch.empty(256 * 1024 * 1024 // 4, device="cuda")
bad = 0
for r in range(rounds):
    for (M, N, K) in shapes:
        for tb in (False, True):
            for epi in (ops.EPI_BF16, ops.EPI_BF16_GELU, ops.EPI_RESID, ops.EPI_BF16_GELUGRAD):
                if only and epi not in only:
                    continue
                A, B, kw = G.make(M, N, K, tb, epi, seed=1000 * r + M + N + K)
                ref, ref2 = G.reference(A, B, kw, tb, epi)
                torch.cuda.synchronize()
                with torch.cuda.stream(side):      # memory traffic beside the kernel
                    junk.mul_(1.0001)
                c, c2 = G.run(A, B, kw, M, N, K, tb, epi, 10)
                torch.cuda.synchronize()
                e = ((c.float() - ref).abs().max() / ref.abs().max()).item() if torch.isfinite(c.float()).all() else float("inf")
                e2 = ((c2.float() - ref2).abs().max() / ref2.abs().max()).item() if ref2 is not None else 0.0
                if not (e < 1.2e-2 and e2 < 1.2e-2):
                    bad += 1
                    print(f"BAD round {r} M={M} N={N} K={K} tb={int(tb)} epi={epi}: {e:.2e} {e2:.2e}", flush=True)
print("rounds", rounds, "FAILURES", bad)
sys.exit(1 if bad else 0)
