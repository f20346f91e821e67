"""MI355X-native DINOv2 training step behind lightly-train's Method / ModelWrapper API.

Layout:
  csrc/      hand-written gfx950 HIP kernels + the C ABI (include/lt_amd.h) -> lib/liblt_amd.so
  _lib.py    ctypes binding of the C ABI (fails loudly when the library is missing)
  ops.py     torch-tensor front-ends of the C ABI entry points (device pointers + current stream)
  vit.py     ViT backbone forward/backward on those ops (mirrors DinoVisionTransformer / ModelWrapper)
  dinov2.py  the DINOv2 Method: teacher/student step, heads, losses, AdamW, EMA (mirrors LT/_methods/dinov2)
"""
import os as _os

# The step runs on six HIP streams (null, teacher, side, local-crop chain, gradient reduce, RCCL fence).  The HIP runtime folds streams
# onto GPU_MAX_HW_QUEUES hardware queues (default 4); two streams on one queue execute IN ORDER, so a weight-gradient GEMM that waits for
# the other backward chain then also holds back the chain that shares its queue: one run in four of the default `bench.py` lost 3-10 ms to
# that with the joint weight gradients (profiles/r04_ab_joint_wgrad.log), none with a queue per stream.  Read by the runtime when it
# initialises (first device call), so importing this package first is enough; an explicit setting of the caller wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

__version__ = "0.1.0"
