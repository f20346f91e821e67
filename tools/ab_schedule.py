"""Same-process A/B of whole stream schedules of the default bench step (ViT-B/16, batch 128, 2 x 224^2 + 8 x 98^2, K = 65 536): the named
presets alternate step by step, each step timed on its own (synchronize + perf_counter) -- the round-4 verdict's "corner nobody measured":
fewer streams with denser kernels against the shipped five-stream schedule.

  python tools/ab_schedule.py [--steps 12] [preset ...]

Presets (attributes of the DINOv2 method object):
  five     shipped: teacher || student-global || student-local forward; two interleaved dgrad chains + the weight-gradient stream
  bwd2     forward as shipped; backward = ONE dgrad chain (local crops, then global crops) + the weight-gradient stream (separate wgrads)
  two      two streams throughout: forward teacher || (global, then local crops on the main stream); backward as `bwd2`
  fwdmain  forward entirely on the main stream (teacher, global, local one after the other); backward as shipped
  tfirst   the teacher's forward on the main stream ahead of the student's (global || local); backward as shipped
  one      every launch on one stream
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

PRESETS = {
    "five": dict(overlap_streams=True, two_bwd_chains=True, fwd_local_stream=1, fwd_teacher_stream=1),
    "bwd2": dict(overlap_streams=True, two_bwd_chains=False, fwd_local_stream=1, fwd_teacher_stream=1),
    "two": dict(overlap_streams=True, two_bwd_chains=False, fwd_local_stream=0, fwd_teacher_stream=1),
    "fwdmain": dict(overlap_streams=True, two_bwd_chains=True, fwd_local_stream=0, fwd_teacher_stream=0),
    "tfirst": dict(overlap_streams=True, two_bwd_chains=True, fwd_local_stream=1, fwd_teacher_stream=0),   # teacher alone first, then global || local
    "one": dict(overlap_streams=False, two_bwd_chains=True, fwd_local_stream=1, fwd_teacher_stream=1),
}
ap = argparse.ArgumentParser()
ap.add_argument("presets", nargs="*", default=["five", "bwd2", "two", "one"])
ap.add_argument("--steps", type=int, default=12, help="timed steps per preset")
ap.add_argument("--batch", type=int, default=128)
ap.add_argument("--env", default=None, help="NAME=v1,v2: a per-call environment switch crossed with the presets (e.g. LT_GEMM_1P=0,3 on a library that has the persistent GEMM)")
a = ap.parse_args()
env_name, env_vals = (a.env.split("=")[0], a.env.split("=")[1].split(",")) if a.env else (None, [None])
cells = [(p_, v_) for p_ in a.presets for v_ in env_vals]

dev = torch.device("cuda", 0)
cfg = ViTConfig(patch_size=16, img_size=224, init_values=1e-5, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4.0)
m = DINOv2(cfg, DINOv2Args(output_dim=65536), global_batch_size=a.batch, total_steps=125_000, device=dev, seed=0)
g = torch.Generator().manual_seed(1234)
views = [torch.randn(a.batch, 3, 224, 224, generator=g).to(dev) for _ in range(2)] + [torch.randn(a.batch, 3, 98, 98, generator=g).to(dev) for _ in range(8)]
random.seed(100)


def apply(name: str) -> None:
    for k, v in PRESETS[name].items():
        setattr(m, k, v)


def enter(cell) -> None:
    apply(cell[0])
    if env_name is not None:
        os.environ[env_name] = cell[1]


for cell in cells:      # every cell's buffers exist before the timing starts
    enter(cell)
    for _ in range(2):
        m.train_step(views)
torch.cuda.synchronize()
t = {c: [] for c in cells}
for i in range(a.steps * len(cells)):
    c = cells[i % len(cells)]
    enter(c)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    m.train_step(views)
    torch.cuda.synchronize()
    t[c].append((time.perf_counter() - t0) * 1e3)
for c in cells:
    x = sorted(t[c])
    tag = c[0] if env_name is None else f"{c[0]} {env_name}={c[1]}"
    print(f"{tag:24s}: median {statistics.median(x):.2f} ms  mean {statistics.fmean(x):.2f}  min {x[0]:.2f}  max {x[-1]:.2f}  (n={len(x)})")
