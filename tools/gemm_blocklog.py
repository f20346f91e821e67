"""Per-block timeline of the persistent two-blocks-per-CU GEMM: which CU/XCD each block ran on, its priority ticket,
start/end ticks and tile count (lt_debug_gemm_log)."""
import ctypes, os, sys, collections, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lightly_train_amd
from lightly_train_amd import ops, _lib

T, D = 256 * 197, 768
M, N, K = T, D, D
A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
B = torch.randn(N, K, device="cuda").to(torch.bfloat16)
C = torch.zeros(M, N, device="cuda")
kw = dict(bias=torch.zeros(N, device="cuda"), gamma=torch.ones(N, device="cuda"), resid=torch.zeros(M, N, device="cuda"),
          out2=torch.empty(M, N, device="cuda", dtype=torch.bfloat16))
for fk in (5, 5, 58):
    ops.gemm(A, B, C, M=M, N=N, K=K, epilogue=ops.EPI_RESID, force_kernel=fk, **kw)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (512 * 8))()
assert _lib.load().lt_debug_gemm_log(ctypes.cast(buf, ctypes.c_void_p), 512) == 0
rows = [list(buf[i * 8:(i + 1) * 8]) for i in range(512)]
t0 = min(r[3] for r in rows)
per_cu = collections.defaultdict(list)
for b, r in enumerate(rows):
    hw, xcc = r[0], r[1]
    cu, sh, se = (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
    per_cu[(xcc, se, sh, cu)].append((b, r[2], (r[3] - t0) / 100.0, (r[4] - t0) / 100.0, r[5]))
print("distinct CUs:", len(per_cu), "blocks/CU histogram:", collections.Counter(len(v) for v in per_cu.values()))
print("ticket pairs:", collections.Counter(tuple(sorted(x[1] for x in v)) for v in per_cu.values()))
print("blockIdx&7 == xcc:", sum(1 for b, r in enumerate(rows) if (b & 7) == r[1]), "/ 512")
for k in sorted(per_cu)[:12]:
    print(k, [(b, pr, f"{s:.1f}-{e:.1f}us", n) for b, pr, s, e, n in per_cu[k]])
tiles = collections.Counter()
for v in per_cu.values():
    for b, pr, s, e, n in v:
        tiles[pr] += n
print("tiles by priority:", dict(tiles), "end max", max(r[4] - t0 for r in rows) / 100.0, "us")
