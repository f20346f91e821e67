"""BASELINE config 5 shape check: DINOv2 ViT-L/14 (SwiGLU-fused FFN), 518^2 global crops (1370 tokens) + 8 x 98^2 local crops,
iBOT on.  Runs a few steps at a small per-GPU batch and reports time / HBM use (sizing for 288 GB)."""
import os, random, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lightly_train_amd
from lightly_train_amd.dinov2 import DINOv2, DINOv2Args
from lightly_train_amd.vit import ViTConfig

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
CKPT = len(sys.argv) > 2 and sys.argv[2] == "ckpt"
cfg = ViTConfig(embed_dim=1024, depth=24, num_heads=16, mlp_ratio=4.0, patch_size=14, img_size=518, init_values=1e-5, ffn_layer="swiglufused")
m = DINOv2(cfg, DINOv2Args(), global_batch_size=B, total_steps=1000, device="cuda", seed=0)
m.activation_checkpointing = CKPT
g = torch.Generator().manual_seed(0)
views = [torch.randn(B, 3, 518, 518, generator=g).cuda() for _ in range(2)] + [torch.randn(B, 3, 98, 98, generator=g).cuda() for _ in range(8)]
random.seed(0)
for i in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = m.train_step(views)
    torch.cuda.synchronize()
    print(f"step {i}: loss {float(res.loss):.4f}  {1e3 * (time.perf_counter() - t0):.1f} ms  ({B / (time.perf_counter() - t0):.1f} img/s)  "
          f"HBM in use {torch.cuda.memory_allocated() / 2**30:.1f} GiB (workspace {m.ws.nbytes() / 2**30:.1f} GiB)")
