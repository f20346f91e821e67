"""LayerNorm-backward bandwidth vs relative alignment of its three fp32 row streams (x, dres, dx)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lightly_train_amd
from lightly_train_amd import ops
rows, D = 50432, 768
n = rows * D
def alloc(off_floats, dtype=torch.float32):
    big = torch.randn(n + (1 << 22), device="cuda").to(dtype)
    return big[off_floats:off_floats + n].view(rows, D)
w = torch.ones(D, device="cuda"); mean = torch.zeros(rows, device="cuda"); rstd = torch.ones(rows, device="cuda")
dw = torch.zeros(D, device="cuda"); db = torch.zeros(D, device="cuda")
for offs in ((0, 0, 0, 0), (0, 1024, 2048, 512), (0, 4096, 8192, 2048), (0, 64, 128, 32), (0, 192 * 1024, 384 * 1024, 0), (0, 1 << 19, 1 << 20, 1 << 18)):
    x, dres, dx = alloc(offs[0]), alloc(offs[1]), alloc(offs[2])
    dy = alloc(offs[3], torch.bfloat16)
    print("ptr mod 2MiB (KiB):", [(t.data_ptr() % (1 << 21)) // 1024 for t in (x, dres, dx, dy)], end="  ")
    for _ in range(3): ops.layernorm_bwd(x, w, mean, rstd, dy, dres, dx, dw, db, rows, D)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.layernorm_bwd(x, w, mean, rstd, dy, dres, dx, dw, db, rows, D)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    print(f"offsets {offs}: {us:.1f} us  {(n * 14) / us / 1e6:.2f} TB/s")
