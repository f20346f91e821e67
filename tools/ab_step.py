"""Same-process A/B of an environment switch that the library reads per call: the default bench step (ViT-B/16, batch 128,
2 x 224^2 + 8 x 98^2 crops, K = 65 536) is run with the values alternating step by step, each step timed on its own
(synchronize + perf_counter).  Whole-run timings of two processes differ by more than most single-kernel changes move the step
(box-to-box +-1.5 %, and a box drifts by several percent over its first minutes).

  python tools/ab_step.py LT_ATTN_BWD 1 2 [--steps 30]
  python tools/ab_step.py two_bwd_chains 0 1 --attr        # a scheduling switch of the method object instead of an environment variable
"""
import argparse
import os
import random
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lightly_train_amd  # noqa: E402,F401
from lightly_train_amd.dinov2 import DINOv2, DINOv2Args  # noqa: E402
from lightly_train_amd.vit import ViTConfig  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("name")
ap.add_argument("values", nargs="+")
ap.add_argument("--steps", type=int, default=30, help="timed steps per value")
ap.add_argument("--batch", type=int, default=128)
ap.add_argument("--attr", action="store_true", help="NAME is an attribute of the DINOv2 method object (e.g. two_bwd_chains, overlap_streams) and the values are ints")
a = ap.parse_args()

dev = torch.device("cuda", 0)
cfg = ViTConfig(patch_size=16, img_size=224, init_values=1e-5, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4.0)
m = DINOv2(cfg, DINOv2Args(output_dim=65536), global_batch_size=a.batch, total_steps=125_000, device=dev, seed=0)
g = torch.Generator().manual_seed(1234)
views = [torch.randn(a.batch, 3, 224, 224, generator=g).to(dev) for _ in range(2)] + [torch.randn(a.batch, 3, 98, 98, generator=g).to(dev) for _ in range(8)]
random.seed(100)
for _ in range(6):
    m.train_step(views)
torch.cuda.synchronize()
t = {v: [] for v in a.values}
for i in range(a.steps * len(a.values)):
    v = a.values[i % len(a.values)]
    if a.attr:   # dotted paths reach into sub-objects: s_head.pad_wgrad_rows; several attributes at once separated by commas
        for name in a.name.split(","):
            obj = m
            *path, leaf = name.split(".")
            for part in path:
                obj = getattr(obj, part)
            setattr(obj, leaf, type(getattr(obj, leaf))(int(v)))
    else:
        os.environ[a.name] = v
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    m.train_step(views)
    torch.cuda.synchronize()
    t[v].append((time.perf_counter() - t0) * 1e3)
print("per-step ms in launch order:", " ".join(f"{a.values[i % len(a.values)]}:{t[a.values[i % len(a.values)]][i // len(a.values)]:.1f}" for i in range(a.steps * len(a.values))))
for v in a.values:
    x = sorted(t[v])
    print(f"{a.name}={v}: median {statistics.median(x):.2f} ms  mean {statistics.fmean(x):.2f}  min {x[0]:.2f}  max {x[-1]:.2f}  (n={len(x)})")
