import os, sys, torch, ctypes
sys.path.insert(0, "/root/repo")
import lightly_train_amd
from lightly_train_amd import ops, _lib
lib = _lib.load()
rows, D = 50432, 768
x = torch.randn(rows, D, device="cuda"); w = torch.ones(D, device="cuda")
mean = torch.zeros(rows, device="cuda"); rstd = torch.ones(rows, device="cuda")
dy = torch.randn(rows, D, device="cuda").to(torch.bfloat16); dres = torch.randn(rows, D, device="cuda")
dx = torch.empty_like(x); dw = torch.zeros(D, device="cuda"); db = torch.zeros(D, device="cuda")
wsb = torch.empty(2048 * 2 * D, device="cuda")
for tag, wsx in (("atomics", None), ("partials", wsb)):
    for _ in range(3): ops.layernorm_bwd(x, w, mean, rstd, dy, dres, dx, dw, db, rows, D, ws=wsx)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.layernorm_bwd(x, w, mean, rstd, dy, dres, dx, dw, db, rows, D, ws=wsx)
    e1.record(); torch.cuda.synchronize()
    print(tag, "%.1f us" % (e0.elapsed_time(e1) / 10 * 1e3))
