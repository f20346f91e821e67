"""Does the host run ahead of the device across steps (so that a scheduler hiccup of the launch thread is absorbed by queued work), or does
something in a step wait for the device?  N default bench steps WITHOUT a synchronize in between: host time at the end of each step's enqueue
against the device time of the step's last kernel (event), plus the wall time of every tensor.to(device) / .item() / float() call that took
longer than 2 ms (a host-blocking transfer shows up here with the device's backlog as its duration).

  python tools/host_ahead_probe.py [--steps 30]
"""
import argparse
import os
import random
import sys
import time
import traceback

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lightly_train_amd  # noqa: E402,F401
from lightly_train_amd.dinov2 import DINOv2, DINOv2Args  # noqa: E402
from lightly_train_amd.vit import ViTConfig  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=30)
a = ap.parse_args()
dev = torch.device("cuda", 0)
cfg = ViTConfig(patch_size=16, img_size=224, init_values=1e-5, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4.0)
m = DINOv2(cfg, DINOv2Args(output_dim=65536), global_batch_size=128, total_steps=125_000, device=dev, seed=0)
g = torch.Generator().manual_seed(1234)
views = [torch.randn(128, 3, 224, 224, generator=g).to(dev) for _ in range(2)] + [torch.randn(128, 3, 98, 98, generator=g).to(dev) for _ in range(8)]
random.seed(100)
for _ in range(6):
    m.train_step(views)
torch.cuda.synchronize()

slow = []
orig_to = torch.Tensor.to


def timed_to(self, *args, **kw):
    t0 = time.perf_counter()
    r = orig_to(self, *args, **kw)
    dt = (time.perf_counter() - t0) * 1e3
    if dt > 2.0:
        fr = [f for f in traceback.extract_stack(limit=6)[:-1] if "lightly-train_amd" in f.filename]
        slow.append((round(dt, 1), "to", tuple(self.shape), f"{os.path.basename(fr[-1].filename)}:{fr[-1].lineno}" if fr else "?"))
    return r


torch.Tensor.to = timed_to
t_start = time.perf_counter()
ev0 = torch.cuda.Event(enable_timing=True); ev0.record()
host_done, evs = [], []
for i in range(a.steps):
    m.train_step(views)
    e = torch.cuda.Event(enable_timing=True); e.record(); evs.append(e)
    host_done.append((time.perf_counter() - t_start) * 1e3)
torch.cuda.synchronize()
torch.Tensor.to = orig_to
dev_done = [ev0.elapsed_time(e) for e in evs]
lead = [dev_done[i] - host_done[i] for i in range(a.steps)]
print("env:", {k: v for k, v in os.environ.items() if k.startswith(("HSA_", "ROC_", "GPU_", "AMD_"))})
print(f"total {dev_done[-1]:.1f} ms for {a.steps} steps = {dev_done[-1] / a.steps:.2f} ms/step; host lead: mean {sum(lead) / len(lead):.0f} ms  min {min(lead):.0f}  max {max(lead):.0f}")
print("lead per step:", " ".join(f"{x:.0f}" for x in lead))
print("slow host calls (> 2 ms):", slow[:40])
