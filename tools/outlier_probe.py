"""Where do the occasional 150-300 ms steps come from?  (tools/ab_step.py: median 88 ms, but 2-3 steps in 50 take 130-320 ms, in rounds 2 and 3
already; bench.py's contract times K consecutive steps, so each of them costs the headline ~1 %.)  The default bench step, N steps, every
step split into its host phases (enqueue of forward + backward, of the optimizer, of the EMA; the wait for the device) with the garbage
collector's runs logged beside them; `--gc-off` freezes / disables the collector after warm-up, `--sync-each` waits for the device after each
phase (host-side stalls then show up in the phase that caused them).

  python tools/outlier_probe.py [--steps 150] [--gc-off]
"""
import argparse
import gc
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
ap.add_argument("--steps", type=int, default=150)
ap.add_argument("--batch", type=int, default=128)
ap.add_argument("--gc-off", action="store_true")
a = ap.parse_args()

dev = torch.device("cuda", 0)
cfg = ViTConfig(patch_size=16, img_size=224, init_values=1e-5, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4.0)
m = DINOv2(cfg, DINOv2Args(output_dim=65536), global_batch_size=a.batch, total_steps=125_000, device=dev, seed=0)
g = torch.Generator().manual_seed(1234)
views = [torch.randn(a.batch, 3, 224, 224, generator=g).to(dev) for _ in range(2)] + [torch.randn(a.batch, 3, 98, 98, generator=g).to(dev) for _ in range(8)]
random.seed(100)
for _ in range(8):
    m.train_step(views)
torch.cuda.synchronize()
gc_log = []
step_now = [0]
t_gc = [0.0]


def on_gc(phase, info):
    if phase == "start":
        t_gc[0] = time.perf_counter()
    else:
        gc_log.append((step_now[0], info["generation"], (time.perf_counter() - t_gc[0]) * 1e3, info.get("collected", 0)))


gc.callbacks.append(on_gc)
if a.gc_off:
    gc.collect()
    gc.freeze()
    gc.disable()
rows = []
mem0 = torch.cuda.memory_stats()
for i in range(a.steps):
    step_now[0] = i
    t0 = time.perf_counter()
    m.training_step_impl({"views": views}, 0)
    t1 = time.perf_counter()
    m.optimizer_step()
    t2 = time.perf_counter()
    m.on_train_batch_end()
    t3 = time.perf_counter()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    rows.append(((t4 - t0) * 1e3, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, int(m._last["M"])))
mem1 = torch.cuda.memory_stats()
tot = [r[0] for r in rows]
med = statistics.median(tot)
print(f"gc {'off' if a.gc_off else 'on'}: steps {a.steps}  median {med:.2f} ms  mean {statistics.fmean(tot):.2f}  max {max(tot):.1f}  "
      f"host enqueue median {statistics.median(r[1] for r in rows):.1f} ms")
print("allocator: cudaMalloc calls during the run", mem1.get("num_device_alloc", 0) - mem0.get("num_device_alloc", 0), " frees",
      mem1.get("num_device_free", 0) - mem0.get("num_device_free", 0), " alloc retries", mem1.get("num_alloc_retries", 0) - mem0.get("num_alloc_retries", 0))
print("gc runs:", [(s, gen, round(ms, 1), n) for s, gen, ms, n in gc_log][:40])
for i, r in enumerate(rows):
    if r[0] > 1.15 * med:
        g_here = [(gen, round(ms, 1)) for s, gen, ms, _ in gc_log if s == i]
        print(f"  step {i:3d}: total {r[0]:6.1f} ms = fwd+bwd enqueue {r[1]:6.1f} | optimizer {r[2]:5.1f} | ema {r[3]:5.1f} | device wait {r[4]:6.1f}   M={r[5]}  gc={g_here}")
