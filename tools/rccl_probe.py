"""Probe (1-GPU box): the RCCL code path of the data-parallel step with a one-rank communicator.

RCCL refuses two ranks on one device, so the N > 1 tests talk over gloo; this probe initialises the real "nccl" backend
exactly as bench.py does (device_id), pretends world = 2 inside the step so that every collective is issued (async center
all-reduces, Sinkhorn row sums, the per-block gradient all-reduces on their own stream, the coalesced log mean) and checks
that two steps run, finish and leave finite parameters.  With one rank every all-reduce is the identity, so the 1/world
scale makes the numbers meaningless -- only the plumbing is under test.

  python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 tools/rccl_probe.py
"""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lightly_train_amd  # noqa: E402,F401
from lightly_train_amd import parallel  # noqa: E402
from lightly_train_amd.dinov2 import DINOv2, DINOv2Args  # noqa: E402
from lightly_train_amd.vit import ViTConfig  # noqa: E402

dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
parallel.world_size = lambda: 2          # issue every collective although the communicator has one rank
DINOv2.world = property(lambda self: 2)

for center in ("softmax", "sinkhorn_knopp"):
    cfg = ViTConfig(patch_size=16, img_size=224, init_values=1e-5, embed_dim=384, depth=12, num_heads=6)
    m = DINOv2(cfg, DINOv2Args(output_dim=4096, center_method=center), global_batch_size=16, total_steps=100, device="cuda")
    g = torch.Generator().manual_seed(0)
    for step in range(2):
        views = [torch.randn(8, 3, 224, 224, generator=g) for _ in range(2)] + [torch.randn(8, 3, 96, 96, generator=g) for _ in range(8)]
        res = m.training_step_impl({"views": views}, step)
        early = sum(b - a for a, b in m._grad_sync.covered)
        logs = m.synced_logs(res)
        m.optimizer_step()
        m.on_train_batch_end()
        torch.cuda.synchronize()
        assert torch.isfinite(res.loss) and torch.isfinite(m.student.data).all()
        print(f"{center} step {step}: loss {float(res.loss):.4f} (synced {float(logs['train_loss']):.4f}), "
              f"{early / m.student.numel:.1%} of the gradient buffer in flight before backward ended, handles left {len(m._grad_sync.handles)}", flush=True)
dist.barrier()
dist.destroy_process_group()
print("rccl probe ok")
