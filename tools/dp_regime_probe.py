"""The stochastic-depth regime (the reference's default for ViT-B: drop_path_rate 0.2, batch-subset form) against the default step, bench geometry:
launch-thread time per step (device idle at its start), per-step wall time with a device sync after every step, and the un-synced average.

  python tools/dp_regime_probe.py [drop_path] [single]"""
import os, random, sys, time
os.environ.setdefault("HSA_KERNARG_POOL_SIZE", str(32 << 20))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lightly_train_amd  # noqa: F401
from lightly_train_amd.dinov2 import DINOv2, DINOv2Args
from lightly_train_amd.vit import ViTConfig

DROP = float(sys.argv[1]) if len(sys.argv) > 1 else 0.2
SINGLE = len(sys.argv) > 2 and sys.argv[2] == "single"
dev = torch.device("cuda", 0)
cfg = ViTConfig(patch_size=16, img_size=224, init_values=1e-5, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4.0, drop_path_rate=DROP)
m = DINOv2(cfg, DINOv2Args(output_dim=65536), global_batch_size=128, total_steps=125_000, device=dev, seed=0)
if SINGLE:
    m.overlap_streams = False
g = torch.Generator().manual_seed(1234)
views = [torch.randn(128, 3, 224, 224, generator=g).to(dev) for _ in range(2)] + [torch.randn(128, 3, 98, 98, generator=g).to(dev) for _ in range(8)]
random.seed(100)
for _ in range(6):
    m.train_step(views)
torch.cuda.synchronize()
host, wall = [], []
for _ in range(10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m.train_step(views)
    t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    host.append((t1 - t0) * 1e3); wall.append((t2 - t0) * 1e3)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(12):
    m.train_step(views)
torch.cuda.synchronize()
free = (time.perf_counter() - t0) / 12 * 1e3
print(f"drop_path {DROP} {'single-stream' if SINGLE else 'five streams'}: launch thread {sorted(host)[5]:.1f} ms (min {min(host):.1f} max {max(host):.1f}) | "
      f"synced step {sorted(wall)[5]:.1f} ms (min {min(wall):.1f} max {max(wall):.1f}) | un-synced average {free:.1f} ms")
