"""Launch plans against eager launches in ONE process (box-to-box and minute-to-minute drift is larger than the effect): two method objects
of the same model, one with the plans off, alternating blocks of un-synced steps; medians of the per-step block times.

  python tools/plan_ab_probe.py [vit_small|vit_base] [blocks] [steps per block]"""
import os, random, statistics, sys, time
os.environ.setdefault("HSA_KERNARG_POOL_SIZE", str(32 << 20))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lightly_train_amd  # noqa: F401
from lightly_train_amd.dinov2 import DINOv2, DINOv2Args
from lightly_train_amd.vit import ViTConfig

ARCH = {"vit_small": (384, 6), "vit_base": (768, 12)}[sys.argv[1] if len(sys.argv) > 1 else "vit_small"]
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 10
NS = int(sys.argv[3]) if len(sys.argv) > 3 else 6
dev = torch.device("cuda", 0)
cfg = ViTConfig(patch_size=16, img_size=224, init_values=1e-5, embed_dim=ARCH[0], depth=12, num_heads=ARCH[1], mlp_ratio=4.0)
g = torch.Generator().manual_seed(1234)
views = [torch.randn(128, 3, 224, 224, generator=g).to(dev) for _ in range(2)] + [torch.randn(128, 3, 98, 98, generator=g).to(dev) for _ in range(8)]
m = DINOv2(cfg, DINOv2Args(output_dim=65536), global_batch_size=128, total_steps=125_000, device=dev, seed=0)
random.seed(100)
for _ in range(6):
    m.train_step(views)        # plans logged on the third step, replayed from the fourth
torch.cuda.synchronize()
saved = {"g": m._bwd_graph}


def select(fwd, bwd, fresh):
    """ONE method object (one set of streams): the variants switch its flags; the backward's plan is put aside while it is off (the eager path
    clears it)."""
    m.s_vit.plan_forward = m.t_vit.plan_forward = fwd
    if bwd and not m.plan_backward:
        m._bwd_graph = saved["g"]
    if not bwd and m.plan_backward:
        saved["g"] = m._bwd_graph
        m._bwd_graph = {}
    m.plan_backward = int(bwd)


variants = {"eager": (False, False, True), "forward plans": (True, False, True), "forward + backward plans": (True, True, True)}
t = {k: [] for k in variants}
for _ in range(NB):
    for k, v in variants.items():
        select(*v)
        m.train_step(views)      # (one untimed step after the switch)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(NS):
            m.train_step(views)
        torch.cuda.synchronize()
        t[k].append((time.perf_counter() - t0) / NS * 1e3)
select(True, True, True)
for k in variants:
    print(f"{sys.argv[1] if len(sys.argv) > 1 else 'vit_small'} {k:52s}: median {statistics.median(t[k]):6.2f} ms/step  min {min(t[k]):6.2f}  max {max(t[k]):6.2f}")
print("backward replays", m._bwd_graph.get("replays", 0), " forward replays", sum(e["replays"] for eng in (m.s_vit, m.t_vit) for e in eng._fwd_plans.values()))
