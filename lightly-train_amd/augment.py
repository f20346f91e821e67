"""GPU multi-crop augmentation: the reference's `DINOTransform` / `ViewTransform` pipeline (LT/_methods/dino/dino_transform.py:129-202,
LT/_transforms/view_transform.py:133-215, arguments of LT/_methods/dinov2/dinov2_transform.py) producing the 2 global + N local
views of a whole batch directly in HBM.

The reference applies, per view, in this order: RandomResizedCrop(size, scale, ratio (3/4, 4/3), cv2.INTER_AREA) -> HorizontalFlip(p)
-> ColorJitter(p; strength * (brightness, contrast, saturation, hue)) -> ToGray(p) -> GaussianBlur(p, sigma range) -> Solarize(p,
threshold) -> Normalize(mean, std) -> ToTensorV2, with the per-view probabilities of `DINOTransformArgs` (global view 0: blur p = 1;
global view 1: blur p = 0.1, solarize p = 0.2; local views: blur p = 0.5, scale (0.05, 0.32) at 98^2 for DINOv2).  It does so in
albumentations / OpenCV on CPU workers.  Here the random PARAMETERS of every (image, view) pair are drawn on the host in one
vectorised numpy pass (same distributions; albumentations' own RNG stream cannot be reproduced without the library) and three HIP
kernels per view size (csrc/augment.hip) do the pixel work for the whole batch."""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
from torch import Tensor

from . import _lib

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)

CROP_DT = np.dtype([("src_off", "<i8"), ("H", "<i4"), ("W", "<i4"), ("x0", "<f4"), ("y0", "<f4"), ("cw", "<f4"), ("ch", "<f4"), ("flip", "<i4")],
                   align=True)
COLOR_DT = np.dtype([("apply", "<i4"), ("order", "<i4"), ("fb", "<f4"), ("fc", "<f4"), ("fs", "<f4"), ("fh", "<f4"), ("gray", "<i4")])
FINISH_DT = np.dtype([("sigma", "<f4"), ("solarize", "<i4"), ("threshold", "<f4")])
assert CROP_DT.itemsize == 40 and COLOR_DT.itemsize == 28 and FINISH_DT.itemsize == 12


@dataclass
class ViewSpec:
    """One entry of DINOTransform.transforms (a `ViewTransformArgs`)."""
    size: int
    scale: Tuple[float, float]
    ratio: Tuple[float, float] = (3.0 / 4.0, 4.0 / 3.0)
    hflip_prob: float = 0.5
    jitter_prob: float = 0.8
    brightness: float = 0.4     # strength 0.5 * (0.8, 0.8, 0.4, 0.2) of DINOColorJitterArgs
    contrast: float = 0.4
    saturation: float = 0.2
    hue: float = 0.1
    gray_prob: float = 0.2
    blur_prob: float = 1.0
    blur_sigma: Tuple[float, float] = (0.1, 2.0)
    solarize_prob: float = 0.0
    solarize_threshold: float = 0.5


def dinov2_view_specs(global_size: int = 224, local_size: int = 98, n_local: int = 8) -> List[ViewSpec]:
    """DINOv2ViTTransformArgs defaults (dinov2_transform.py: global scale (0.32, 1.0), local (0.05, 0.32) at 98^2, 8 local views) on
    top of DINOTransformArgs (dino_transform.py:36-112)."""
    return [ViewSpec(global_size, (0.32, 1.0), blur_prob=1.0),
            ViewSpec(global_size, (0.32, 1.0), blur_prob=0.1, solarize_prob=0.2)] + \
           [ViewSpec(local_size, (0.05, 0.32), blur_prob=0.5) for _ in range(n_local)]


def sample_crop_boxes(H: np.ndarray, W: np.ndarray, scale: Tuple[float, float], ratio: Tuple[float, float], rng: np.random.Generator) -> np.ndarray:
    """RandomResizedCrop's box (torchvision `get_params`, which albumentations follows): up to 10 attempts of
    area ~ U(scale) * H * W, log-ratio ~ U(log r0, log r1), w = round(sqrt(area * r)), h = round(sqrt(area / r)), accepted when the box
    fits; else the central crop with the ratio clamped to the range.  Vectorised over images.  Returns float32 [n, 4] = x0, y0, w, h."""
    n = H.shape[0]
    out = np.zeros((n, 4), np.float32)
    done = np.zeros(n, bool)
    area_img = (H * W).astype(np.float64)
    lr0, lr1 = math.log(ratio[0]), math.log(ratio[1])
    for _ in range(10):
        area = area_img * rng.uniform(scale[0], scale[1], n)
        r = np.exp(rng.uniform(lr0, lr1, n))
        w = np.rint(np.sqrt(area * r)).astype(np.int64)
        h = np.rint(np.sqrt(area / r)).astype(np.int64)
        ok = (~done) & (w > 0) & (w <= W) & (h > 0) & (h <= H)
        y0 = np.floor(rng.uniform(0, 1, n) * (H - h + 1)).astype(np.int64)
        x0 = np.floor(rng.uniform(0, 1, n) * (W - w + 1)).astype(np.int64)
        out[ok] = np.stack([x0, y0, w, h], 1)[ok]
        done |= ok
    if not done.all():   # fallback: central crop
        in_ratio = W / H
        w = np.where(in_ratio < ratio[0], W, np.where(in_ratio > ratio[1], np.rint(H * ratio[1]), W)).astype(np.int64)
        h = np.where(in_ratio < ratio[0], np.rint(W / ratio[0]), np.where(in_ratio > ratio[1], H, H)).astype(np.int64)
        fb = np.stack([(W - w) // 2, (H - h) // 2, w, h], 1).astype(np.float32)
        out[~done] = fb[~done]
    return out


_PERMS = [(a, b, c, d) for a in range(4) for b in range(4) for c in range(4) for d in range(4) if len({a, b, c, d}) == 4]


def sample_view_params(spec: ViewSpec, H: np.ndarray, W: np.ndarray, src_off: np.ndarray, rng: np.random.Generator):
    """The three record arrays (crop / colour / finish) of one view of every image of the batch."""
    n = H.shape[0]
    crop = np.zeros(n, CROP_DT)
    box = sample_crop_boxes(H, W, spec.scale, spec.ratio, rng)
    crop["src_off"], crop["H"], crop["W"] = src_off, H, W
    crop["x0"], crop["y0"], crop["cw"], crop["ch"] = box[:, 0], box[:, 1], box[:, 2], box[:, 3]
    crop["flip"] = rng.uniform(0, 1, n) < spec.hflip_prob
    col = np.zeros(n, COLOR_DT)
    col["apply"] = rng.uniform(0, 1, n) < spec.jitter_prob
    perm = np.array(_PERMS, np.int32)[rng.integers(0, 24, n)]
    col["order"] = perm[:, 0] | (perm[:, 1] << 2) | (perm[:, 2] << 4) | (perm[:, 3] << 6)
    col["fb"] = rng.uniform(max(0.0, 1 - spec.brightness), 1 + spec.brightness, n)
    col["fc"] = rng.uniform(max(0.0, 1 - spec.contrast), 1 + spec.contrast, n)
    col["fs"] = rng.uniform(max(0.0, 1 - spec.saturation), 1 + spec.saturation, n)
    col["fh"] = rng.uniform(-spec.hue, spec.hue, n)
    col["gray"] = rng.uniform(0, 1, n) < spec.gray_prob
    fin = np.zeros(n, FINISH_DT)
    blur = rng.uniform(0, 1, n) < spec.blur_prob
    fin["sigma"] = np.where(blur, rng.uniform(spec.blur_sigma[0], spec.blur_sigma[1], n), 0.0)
    fin["solarize"] = rng.uniform(0, 1, n) < spec.solarize_prob
    fin["threshold"] = spec.solarize_threshold
    return crop, col, fin


def _dev(arr: np.ndarray, device: torch.device) -> Tensor:
    return torch.from_numpy(arr.view(np.uint8).reshape(-1)).to(device, non_blocking=True)


class GPUMultiCrop:
    """`DINOTransform.__call__` for a whole batch: images (decoded uint8 HWC, packed in one device buffer) -> list of view tensors
    f32 [B, 3, S, S], in the reference's view order [global 0, global 1, local 0 .. N-1]."""

    def __init__(self, specs: Optional[Sequence[ViewSpec]] = None, mean: Sequence[float] = IMAGENET_MEAN, std: Sequence[float] = IMAGENET_STD,
                 seed: int = 0, device: str | torch.device = "cuda") -> None:
        self.specs = list(specs) if specs is not None else dinov2_view_specs()
        self.rng = np.random.default_rng(seed)
        self.device = torch.device(device)
        self._mean = (C.c_float * 3)(*mean)
        self._std = (C.c_float * 3)(*std)
        self._scratch: Dict[Tuple[int, int], Tensor] = {}

    @staticmethod
    def pack(images: Sequence[Tensor], device: str | torch.device = "cuda") -> Tuple[Tensor, np.ndarray, np.ndarray, np.ndarray]:
        """uint8 [H, W, 3] images (any sizes) -> (one packed device buffer, H[n], W[n], byte offsets[n])."""
        H = np.array([im.shape[0] for im in images], np.int32)
        W = np.array([im.shape[1] for im in images], np.int32)
        sizes = H.astype(np.int64) * W * 3
        off = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
        buf = torch.empty(int(sizes.sum()), dtype=torch.uint8, device=device)
        for im, o, s in zip(images, off, sizes):
            assert im.dtype == torch.uint8 and im.dim() == 3 and im.shape[2] == 3
            buf[int(o):int(o + s)].copy_(im.reshape(-1), non_blocking=True)
        return buf, H, W, off

    def __call__(self, packed: Tensor, H: np.ndarray, W: np.ndarray, src_off: np.ndarray, params: Optional[list] = None) -> List[Tensor]:
        lib = _lib.load()
        st = torch.cuda.current_stream().cuda_stream
        B = H.shape[0]
        if params is None:
            params = [sample_view_params(sp, H, W, src_off, self.rng) for sp in self.specs]
        # group the views by output size: one launch of each kernel per size
        groups: Dict[int, List[int]] = {}
        for vi, sp in enumerate(self.specs):
            groups.setdefault(sp.size, []).append(vi)
        views: List[Optional[Tensor]] = [None] * len(self.specs)
        for S, vis in groups.items():
            n = B * len(vis)
            crop = np.concatenate([params[v][0] for v in vis])
            col = np.concatenate([params[v][1] for v in vis])
            fin = np.concatenate([params[v][2] for v in vis])
            d_crop, d_col, d_fin = _dev(crop, self.device), _dev(col, self.device), _dev(fin, self.device)
            tmp = self._scratch.get((n, S))
            if tmp is None:
                tmp = torch.empty(n, 3, S, S, dtype=torch.float32, device=self.device)
                self._scratch[(n, S)] = tmp
            out = torch.empty(n, 3, S, S, dtype=torch.float32, device=self.device)
            _lib.check(lib.lt_aug_crop_resize(packed.data_ptr(), d_crop.data_ptr(), tmp.data_ptr(), n, S, st), "lt_aug_crop_resize")
            _lib.check(lib.lt_aug_color(tmp.data_ptr(), d_col.data_ptr(), n, S, st), "lt_aug_color")
            _lib.check(lib.lt_aug_finish(tmp.data_ptr(), d_fin.data_ptr(), out.data_ptr(), n, S, C.cast(self._mean, C.c_void_p),
                                         C.cast(self._std, C.c_void_p), st), "lt_aug_finish")
            for k, v in enumerate(vis):
                views[v] = out[k * B:(k + 1) * B]
        return views  # type: ignore[return-value]
