"""Would ONE student pass over global + local crops pay?  Isolated GEMM times of the step's token shapes at the global (50 432), local
(51 200) and merged (101 632) row counts, and of the weight-gradient GEMMs at the single / merged contraction lengths."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lightly_train_amd
from lightly_train_amd import ops
sys.argv = [sys.argv[0], "0", "fwd-only"]
import importlib.util
spec = importlib.util.spec_from_file_location("gb", os.path.join(ROOT, "tools", "gemm_bench.py"))
gb = importlib.util.module_from_spec(spec)
try:
    spec.loader.exec_module(gb)
except SystemExit:
    pass
D = 768
Tg, Tl = 256 * 197, 1024 * 50
WS = torch.empty(64 * 1024 * 1024, device="cuda")
for T in (Tg, Tl, Tg + Tl):
    print(f"---- rows {T}")
    gb.bench("qkv fwd", T, 3 * D, D, False, False, ops.EPI_BF16)
    gb.bench("proj fwd resid", T, D, D, False, False, ops.EPI_RESID)
    gb.bench("fc1 fwd gelu", T, 4 * D, D, False, False, ops.EPI_BF16_GELU)
    gb.bench("fc2 fwd resid", T, D, 4 * D, False, False, ops.EPI_RESID)
    gb.bench("fc2 dgrad gelugrad", T, 4 * D, D, False, True, ops.EPI_BF16_GELUGRAD)
    gb.bench("fc1 dgrad", T, D, 4 * D, False, True, ops.EPI_BF16)
    gb.bench("qkv dgrad", T, D, 3 * D, False, True, ops.EPI_BF16)
    gb.bench("proj dgrad", T, D, D, False, True, ops.EPI_BF16)
    for nm, mm, nn in (("fc1 wgrad", 4 * D, D), ("fc2 wgrad", D, 4 * D), ("qkv wgrad", 3 * D, D), ("proj wgrad", D, D)):
        gb.bench(nm + " slab", mm, nn, T, True, True, ops.EPI_F32_ACCUM, split=0, ws=WS)
