"""Same-process A/B of two builds of liblt_amd.so: the default bench step (ViT-B/16, batch 128, 2 x 224^2 + 8 x 98^2 crops,
K = 65 536) runs with the two libraries alternating step by step (the ctypes handle the launch wrappers use is swapped between
steps), each step timed on its own.  Both builds must export the same ABI; weights, buffers and streams are shared.

  python tools/ab_lib.py lightly-train_amd/lib/liblt_amd_variant.so [--steps 30] [--model vit_small]
"""
import argparse
import ctypes as C
import os
import random
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lightly_train_amd  # noqa: E402,F401
from lightly_train_amd import _lib  # noqa: E402
from lightly_train_amd.dinov2 import DINOv2, DINOv2Args  # noqa: E402
from lightly_train_amd.vit import ViTConfig  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("variant")
ap.add_argument("--steps", type=int, default=30, help="timed steps per library")
ap.add_argument("--batch", type=int, default=128)
ap.add_argument("--model", default="vit_base", choices=["vit_base", "vit_small"])
a = ap.parse_args()


def bind(path: str) -> C.CDLL:
    lib = C.CDLL(os.path.abspath(path))
    lib.lt_last_error.restype = C.c_char_p
    lib.lt_last_error.argtypes = []
    lib.lt_attention_bwd_ws_floats.restype = C.c_int64
    lib.lt_attention_bwd_ws_floats.argtypes = [C.c_int] * 4
    lib.lt_batchnorm_ws_floats.restype = C.c_int64
    lib.lt_batchnorm_ws_floats.argtypes = [C.c_int]
    for name, argtypes in _lib.SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = C.c_int
    assert lib.lt_abi_version() == _lib.ABI_VERSION
    return lib


libs = {"shipped": _lib.load(), "variant": bind(a.variant)}
dev = torch.device("cuda", 0)
arch = dict(vit_base=dict(embed_dim=768, depth=12, num_heads=12), vit_small=dict(embed_dim=384, depth=12, num_heads=6))[a.model]
cfg = ViTConfig(patch_size=16, img_size=224, init_values=1e-5, mlp_ratio=4.0, **arch)
m = DINOv2(cfg, DINOv2Args(output_dim=65536), global_batch_size=a.batch, total_steps=125_000, device=dev, seed=0)
g = torch.Generator().manual_seed(1234)
views = [torch.randn(a.batch, 3, 224, 224, generator=g).to(dev) for _ in range(2)] + [torch.randn(a.batch, 3, 98, 98, generator=g).to(dev) for _ in range(8)]
random.seed(100)
for k in ("shipped", "variant") * 3:
    _lib._lib = libs[k]
    m.train_step(views)
torch.cuda.synchronize()
t = {k: [] for k in libs}
order = ["shipped", "variant"]
for i in range(a.steps * 2):
    k = order[i % 2]
    _lib._lib = libs[k]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = m.train_step(views)
    torch.cuda.synchronize()
    t[k].append((time.perf_counter() - t0) * 1e3)
_lib._lib = libs["shipped"]
for k in order:
    x = sorted(t[k])
    print(f"{k:8s}: median {statistics.median(x):.2f} ms  mean {statistics.fmean(x):.2f}  min {x[0]:.2f}  max {x[-1]:.2f}  (n={len(x)})")
print("final loss", float(res.loss))
