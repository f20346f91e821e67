"""Where do the small copies / fills of a step come from?  One bench-shaped DINOv2 step under torch.profiler: aten::copy_ / fill_ / zero_ calls
grouped by the repo line that issued them."""
import collections
import os
import sys
import traceback

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lightly_train_amd  # noqa: E402,F401
from lightly_train_amd.dinov2 import DINOv2, DINOv2Args  # noqa: E402
from lightly_train_amd.vit import ViTConfig  # noqa: E402

cfg = ViTConfig(embed_dim=768, depth=12, num_heads=12, patch_size=16, img_size=224, init_values=1e-5)
B = 32
m = DINOv2(cfg, DINOv2Args(), global_batch_size=B, total_steps=1000, device="cuda", seed=0)
g = torch.Generator().manual_seed(0)
views = [torch.randn(B, 3, 224, 224, generator=g).cuda() for _ in range(2)] + [torch.randn(B, 3, 98, 98, generator=g).cuda() for _ in range(8)]
for _ in range(2):
    m.train_step(views)
torch.cuda.synchronize()

counts = collections.Counter()
orig = {}


def wrap(name):
    fn = getattr(torch.Tensor, name)
    orig[name] = fn

    def inner(self, *a, **k):
        for fr in reversed(traceback.extract_stack(limit=8)[:-1]):
            if "lightly-train_amd" in fr.filename:
                counts[(name, os.path.basename(fr.filename), fr.lineno, tuple(self.shape) if self.numel() < 1 << 20 else "big")] += 1
                break
        return fn(self, *a, **k)

    setattr(torch.Tensor, name, inner)


for n in ("copy_", "zero_", "fill_", "to", "contiguous", "clone"):
    wrap(n)
m.train_step(views)
torch.cuda.synchronize()
for n, fn in orig.items():
    setattr(torch.Tensor, n, fn)
tot = collections.Counter()
for (name, f, line, shape), c in counts.items():
    tot[(name, f, line)] += c
for (name, f, line), c in tot.most_common(40):
    print(f"{c:5d}  {name:10s} {f}:{line}")
print("total", sum(tot.values()))
