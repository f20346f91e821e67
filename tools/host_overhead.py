"""Diagnostic: host-side (launch) time per step vs GPU time."""
import os, sys, time, random, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lightly_train_amd
from lightly_train_amd.dinov2 import DINOv2, DINOv2Args
from lightly_train_amd.vit import ViTConfig
ARCH = {"vit_small": (384, 6), "vit_base": (768, 12)}[sys.argv[1] if len(sys.argv) > 1 else "vit_base"]
DROP = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0     # python tools/host_overhead.py vit_base 0.2: the batch-subset stochastic-depth regime
cfg = ViTConfig(embed_dim=ARCH[0], depth=12, num_heads=ARCH[1], patch_size=16, img_size=224, init_values=1e-5, drop_path_rate=DROP)
m = DINOv2(cfg, DINOv2Args(), global_batch_size=128, total_steps=125000, device="cuda")
g = torch.Generator().manual_seed(0)
B = 128
views = [torch.randn(B, 3, 224, 224, generator=g).cuda() for _ in range(2)] + [torch.randn(B, 3, 96, 96, generator=g).cuda() for _ in range(8)]
random.seed(0)
for _ in range(5):     # (launch plans are logged on the third step of a geometry and replayed from the fourth)
    m.train_step(views)
torch.cuda.synchronize()
import cProfile, pstats
host = []
for _ in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    m.train_step(views)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append((t1 - t0, t2 - t0))
print("host-only ms, total ms:", [(round(a * 1e3, 1), round(b * 1e3, 1)) for a, b in host])
# mask generation alone
from lightly_train_amd.masking import MaskingGenerator, create_collated_masks
t0 = time.perf_counter()
for _ in range(5):
    gen = MaskingGenerator(input_size=(14, 14), max_num_patches=98)
    create_collated_masks(0.1, 0.5, 128, 256, gen)
print("mask sampling ms:", (time.perf_counter() - t0) / 5 * 1e3)
pr = cProfile.Profile(); pr.enable(); m.train_step(views); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(30)
