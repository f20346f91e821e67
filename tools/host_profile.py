"""Diagnostic: cProfile of N back-to-back steps (no synchronisation between them: the host runs ahead of the device as in bench.py),
sorted by the time spent inside each function itself.   python tools/host_profile.py [--drop-path 0.2] [--steps 6]"""
import argparse, cProfile, os, pstats, random, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lightly_train_amd  # noqa: F401
from lightly_train_amd.dinov2 import DINOv2, DINOv2Args
from lightly_train_amd.vit import ViTConfig
ap = argparse.ArgumentParser()
ap.add_argument("--drop-path", type=float, default=0.0)
ap.add_argument("--steps", type=int, default=6)
a = ap.parse_args()
cfg = ViTConfig(embed_dim=768, depth=12, num_heads=12, patch_size=16, img_size=224, init_values=1e-5, drop_path_rate=a.drop_path)
m = DINOv2(cfg, DINOv2Args(), global_batch_size=128, total_steps=125000, device="cuda")
g = torch.Generator().manual_seed(0)
B = 128
views = [torch.randn(B, 3, 224, 224, generator=g).cuda() for _ in range(2)] + [torch.randn(B, 3, 98, 98, generator=g).cuda() for _ in range(8)]
random.seed(0)
for _ in range(3):
    m.train_step(views)
torch.cuda.synchronize()
t0 = time.perf_counter()
pr = cProfile.Profile(); pr.enable()
for _ in range(a.steps):
    m.train_step(views)
pr.disable()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host {1e3 * (t1 - t0) / a.steps:.1f} ms/step (under cProfile), wall {1e3 * (t2 - t0) / a.steps:.1f} ms/step")
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
