"""What the step would gain if a kernel class cost nothing: the default bench step (ViT-B/16, batch 128, 2 x 224^2 + 8 x 98^2, K = 65 536) with the launches of one
class at a time replaced by no-ops (their outputs stay whatever the previous step left: wrong values, same remaining work), alternating with the unmodified step
in one process.  Under the five-stream schedule a kernel's standalone time says little about what it costs the step; this does.

  python tools/sensitivity_probe.py [--steps 6] [--rounds 3]
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
from lightly_train_amd import ops  # noqa: E402
from lightly_train_amd.dinov2 import DINOv2, DINOv2Args  # noqa: E402
from lightly_train_amd.vit import ViTConfig  # noqa: E402

CLASSES = {
    "baseline": (),
    "LayerNorm forward": ("layernorm_fwd",),
    "LayerNorm backward": ("layernorm_bwd",),
    "attention forward": ("attention_fwd",),
    "attention backward": ("attention_bwd",),
    "teacher statistics + cross-entropy": ("softmax_stats_colsum", "ce_fwd_bwd_logits"),
    "AdamW + EMA + gradient norm": ("adamw_flat", "ema_flat", "sumsq"),
    "row gathers / scatters, LayerScale backward": ("gather_rows", "scatter_add_rows", "layerscale_bwd"),
    "KoLeo": ("koleo_fwd_bwd",),
    "all of the above": ("layernorm_fwd", "layernorm_bwd", "attention_fwd", "attention_bwd", "softmax_stats_colsum", "ce_fwd_bwd_logits", "adamw_flat",
                         "ema_flat", "sumsq", "gather_rows", "scatter_add_rows", "layerscale_bwd", "koleo_fwd_bwd"),
}
ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=6)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--batch", type=int, default=128)
a = ap.parse_args()
dev = torch.device("cuda", 0)
cfg = ViTConfig(patch_size=16, img_size=224, init_values=1e-5, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4.0)
m = DINOv2(cfg, DINOv2Args(output_dim=65536), global_batch_size=a.batch, total_steps=125_000, device=dev, seed=0)
g = torch.Generator().manual_seed(1234)
views = [torch.randn(a.batch, 3, 224, 224, generator=g).to(dev) for _ in range(2)] + [torch.randn(a.batch, 3, 98, 98, generator=g).to(dev) for _ in range(8)]
random.seed(100)
real = {n: getattr(ops, n) for names in CLASSES.values() for n in names}


def noop(*args, **kw):
    return None


def run(n: int) -> float:
    """n steps back to back (the launch thread runs ahead of the device, as in bench.py), one synchronize at the end: ms per step"""
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        m.train_step(views)      # (training step + optimizer step + EMA)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / n


run(6)
times = {k: [] for k in CLASSES}
for it in range(a.rounds):
    for name, fns in CLASSES.items():
        for n in real:
            setattr(ops, n, noop if n in fns else real[n])
        run(1)                       # (the first step after a switch still drains the previous cell's queue)
        times[name].append(run(a.steps))
for n in real:
    setattr(ops, n, real[n])
base = statistics.median(times["baseline"])
print(f"| kernel class replaced by no-ops | ms per step (median of {a.rounds} runs of {a.steps} steps) | gain ms |\n|---|---|---|")
for name in CLASSES:
    med = statistics.median(times[name])
    print(f"| {name} | {med:.2f} | {base - med:+.2f} |")
