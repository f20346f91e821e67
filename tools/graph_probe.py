"""Diagnostic: five steps of a small DINOv2 model with the HIP-graph switches given in the environment (LT_GRAPH_FWD / LT_GRAPH_BWD and the
schedule switches they interact with); prints the loss of every step.  Run per configuration in a fresh process (a failed capture can
take the process down)."""
import os, random, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lightly_train_amd  # noqa: F401
from lightly_train_amd.dinov2 import DINOv2, DINOv2Args
from lightly_train_amd.vit import ViTConfig
B, gsz, lsz = 8, 112, 48
cfg = ViTConfig(embed_dim=384, depth=4, num_heads=6, mlp_ratio=4.0, patch_size=16, img_size=gsz, init_values=1e-2)
m = DINOv2(cfg, DINOv2Args(output_dim=8192, hidden_dim=512, dino_bottleneck_dim=256), global_batch_size=B, total_steps=100, device="cuda", seed=3)
g = torch.Generator().manual_seed(0)
views = [torch.randn(B, 3, gsz, gsz, generator=g) for _ in range(2)] + [torch.randn(B, 3, lsz, lsz, generator=g) for _ in range(4)]
for step in range(5):
    random.seed(100 + step)
    res = m.train_step(views)
    torch.cuda.synchronize()
    print("step", step, float(res.loss), flush=True)
print("OK", {k: os.environ.get(k) for k in ("LT_GRAPH_FWD", "LT_GRAPH_BWD", "LT_JOINT_WGRAD", "LT_DETERMINISTIC")}, flush=True)
