"""MI355X-native DINOv2 training step behind lightly-train's Method / ModelWrapper API.

Layout:
  csrc/      hand-written gfx950 HIP kernels + the C ABI (include/lt_amd.h) -> lib/liblt_amd.so
  _lib.py    ctypes binding of the C ABI (fails loudly when the library is missing)
  ops.py     torch-tensor front-ends of the C ABI entry points (device pointers + current stream)
  vit.py     ViT backbone forward/backward on those ops (mirrors DinoVisionTransformer / ModelWrapper)
  dinov2.py  the DINOv2 Method: teacher/student step, heads, losses, AdamW, EMA (mirrors LT/_methods/dinov2)
"""
__version__ = "0.1.0"
