"""Which calls of the stochastic-depth step block the launch thread?  Every ops.* front-end is wrapped with a wall-clock timer; calls longer than
0.3 ms are listed with their position in the step (un-synced steps: the host runs ahead of the device as in training)."""
import os, random, sys, time
os.environ.setdefault("HSA_KERNARG_POOL_SIZE", str(32 << 20))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lightly_train_amd  # noqa: F401
from lightly_train_amd import ops
from lightly_train_amd.dinov2 import DINOv2, DINOv2Args
from lightly_train_amd.vit import ViTConfig

DROP = float(sys.argv[1]) if len(sys.argv) > 1 else 0.2
dev = torch.device("cuda", 0)
cfg = ViTConfig(patch_size=16, img_size=224, init_values=1e-5, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4.0, drop_path_rate=DROP)
m = DINOv2(cfg, DINOv2Args(output_dim=65536), global_batch_size=128, total_steps=125_000, device=dev, seed=0)
g = torch.Generator().manual_seed(1234)
views = [torch.randn(128, 3, 224, 224, generator=g).to(dev) for _ in range(2)] + [torch.randn(128, 3, 98, 98, generator=g).to(dev) for _ in range(8)]
random.seed(100)
for _ in range(6):
    m.train_step(views)
torch.cuda.synchronize()
log = []
step_no = [0]
def wrap(name):
    fn = getattr(ops, name)
    def w(*a, **k):
        t0 = time.perf_counter()
        r = fn(*a, **k)
        dt = (time.perf_counter() - t0) * 1e3
        if dt > 0.3:
            desc = ""
            if name == "gemm":
                desc = f"M={k.get('M')} N={k.get('N')} K={k.get('K')} epi={k.get('epilogue', 0)} ta={int(k.get('trans_a', False))}"
            elif name == "h2d":
                desc = f"bytes={a[0].numel() * a[0].element_size()}"
            log.append((step_no[0], len(calls), name, dt, desc))
        calls.append(name)
        return r
    setattr(ops, name, w)
calls = []
for n in dir(ops):
    f = getattr(ops, n)
    if callable(f) and not n.startswith("_") and getattr(f, "__module__", "") == ops.__name__ and n not in ("check", "record_plan", "LaunchPlan", "recordable", "require_device"):
        wrap(n)
t_steps = []
for s in range(10):
    step_no[0] = s; calls.clear()
    t0 = time.perf_counter()
    m.train_step(views)
    t_steps.append((time.perf_counter() - t0) * 1e3)
torch.cuda.synchronize()
print("launch-thread ms per step (un-synced):", " ".join(f"{t:.1f}" for t in t_steps), " calls per step:", len(calls))
for e in log:
    print("step %d call #%d %-22s %7.2f ms  %s" % e)
