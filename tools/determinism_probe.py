"""Run the same DINOv2 step several times from one state and list the gradient tensors (and loss terms) that are not bitwise identical
between runs.  usage: python tools/determinism_probe.py [runs]"""
import os
import random
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lightly_train_amd  # noqa: E402,F401
from lightly_train_amd.dinov2 import DINOv2, DINOv2Args  # noqa: E402
from lightly_train_amd.vit import ViTConfig  # noqa: E402


def main() -> None:
    runs = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    cfg = ViTConfig(embed_dim=384, depth=3, num_heads=6, mlp_ratio=4.0, patch_size=16, img_size=224, init_values=1e-2, drop_path_rate=0.1)
    args = DINOv2Args(output_dim=8192, hidden_dim=512, dino_bottleneck_dim=256)
    B = 16
    m = DINOv2(cfg, args, global_batch_size=B, total_steps=100, device="cuda", seed=3)
    if os.environ.get("LT_PROBE_NO_OVERLAP"):
        m.overlap_streams = False
    g = torch.Generator().manual_seed(0)
    views = [torch.randn(B, 3, 224, 224, generator=g) for _ in range(2)] + [torch.randn(B, 3, 96, 96, generator=g) for _ in range(8)]
    m.train_step(views)           # one optimizer step first: non-trivial LayerScale / bias values
    ref = None
    differing = {}
    shown = set()
    for r in range(runs):
        random.seed(11)
        m._drop_gen.manual_seed(5)
        m._pending.clear()
        res = m.training_step_impl({"views": views}, 0)
        torch.cuda.synchronize()
        snap = {n: m.student.g[n].clone() for n in m.student.names}
        if os.environ.get("LT_PROBE_WS"):
            for k_, t_ in m.ws.bufs.items():
                if t_.numel() * t_.element_size() < 256 * 1024 * 1024 and "slabs" not in k_ and "scratch" not in k_:
                    snap["ws:" + k_] = t_.clone()
        snap["__loss_slots"] = m._loss_slots.clone()
        if ref is None:
            ref = snap
            continue
        for n, v in snap.items():
            if n not in ref or v.shape != ref[n].shape:
                continue
            if n.startswith("ws:"):
                if not torch.equal(v.view(torch.uint8), ref[n].view(torch.uint8)):
                    differing[n] = 1.0
                continue
            if not torch.equal(v, ref[n]):
                d = (v - ref[n]).abs().max().item() / (ref[n].abs().max().item() + 1e-30)
                differing[n] = max(differing.get(n, 0.0), d)
                if v.ndim == 2 and n not in shown:
                    shown.add(n)
                    bad = (v != ref[n]).nonzero()
                    print(f"    {n}: {bad.shape[0]} of {v.numel()} elements differ; rows {int(bad[:, 0].min())}..{int(bad[:, 0].max())} "
                          f"({bad[:, 0].unique().numel()} distinct), cols {int(bad[:, 1].min())}..{int(bad[:, 1].max())} ({bad[:, 1].unique().numel()} distinct)")
    from lightly_train_amd import ops
    print(f"{len(differing)} of {len(ref)} tensors differ between {runs} runs; ledger overflows {ops.reduce_overflows()}")
    for n, d in sorted(differing.items()):
        print(f"  {n:60s} {d:.2e}")


if __name__ == "__main__":
    main()
