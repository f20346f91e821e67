"""Torch-free core of the iBOT block-mask sampler (mirrors MaskingGenerator / the sampling half of create_collated_masks,
LT/_methods/dinov2/utils.py:41-152).  RNG is Python's `random` -- the module itself (the reference's global stream) or a private
`random.Random` carrying that stream -- consumed in exactly the reference's order.  Kept free of torch so that the background
producer process (masking.MaskProducer) starts in a fraction of a second."""
from __future__ import annotations

import math
import random
from typing import Any, List, Tuple

import numpy as np


class MaskingGenerator:
    def __init__(self, input_size: int | Tuple[int, int], max_num_patches: int, min_num_patches: int = 4,
                 min_aspect: float = 0.3, max_aspect: float | None = None, rng: Any = random) -> None:
        self.rng = rng   # the `random` module (the reference's global stream) or a private random.Random carrying that stream
        if not isinstance(input_size, tuple):
            input_size = (input_size, input_size)
        self.height, self.width = input_size
        self.num_patches = self.height * self.width
        self.min_num_patches = min_num_patches
        self.max_num_patches = max_num_patches
        max_aspect = max_aspect or 1 / min_aspect
        self.log_aspect_ratio = (math.log(min_aspect), math.log(max_aspect))

    def get_shape(self) -> Tuple[int, int]:
        return self.height, self.width

    def _try_block(self, grid: np.ndarray, budget: int) -> int:
        gained = 0
        for _attempt in range(10):
            area = self.rng.uniform(self.min_num_patches, budget)
            aspect = math.exp(self.rng.uniform(*self.log_aspect_ratio))
            bh = int(round(math.sqrt(area * aspect)))
            bw = int(round(math.sqrt(area / aspect)))
            if bw < self.width and bh < self.height:
                y0 = self.rng.randint(0, self.height - bh)
                x0 = self.rng.randint(0, self.width - bw)
                window = grid[y0:y0 + bh, x0:x0 + bw]
                fresh = bh * bw - int(window.sum())
                if 0 < fresh <= budget:
                    window[...] = True
                    gained += fresh
            if gained > 0:
                break
        return gained

    def __call__(self, num_masking_patches: int = 0) -> np.ndarray:
        grid = np.zeros((self.height, self.width), dtype=bool)
        done = 0
        while done < num_masking_patches:
            budget = min(num_masking_patches - done, self.max_num_patches)
            got = self._try_block(grid, budget)
            if got == 0:
                break
            done += got
        return grid


def sample_mask_grids(mask_ratio_min: float, mask_ratio_max: float, n_masked_crops: int, n_crops: int,
                      mask_generator: MaskingGenerator) -> np.ndarray:
    """bool [n_crops, H, W]: the masks of one step in their final (shuffled) order."""
    n_tokens = mask_generator.num_patches
    edges = np.linspace(mask_ratio_min, mask_ratio_max, n_masked_crops + 1)
    grids: List[np.ndarray] = []
    for i in range(n_masked_crops):
        target = int(n_tokens * mask_generator.rng.uniform(edges[i], edges[i + 1]))
        grids.append(mask_generator(target))
    for _ in range(n_masked_crops, n_crops):
        grids.append(mask_generator(0))
    mask_generator.rng.shuffle(grids)
    return np.stack(grids)


def sample_mask_grids_native(mask_ratio_min: float, mask_ratio_max: float, n_masked_crops: int, n_crops: int,
                             mask_generator: MaskingGenerator) -> np.ndarray:
    """Same result and same consumption of the `random` stream as `sample_mask_grids`, computed by `lt_sample_block_masks`
    (csrc/host_masks.cpp: the sampler in C++ on CPython's own Mersenne-Twister state): ~0.1 ms instead of ~13 ms for 256 crops."""
    import ctypes as C

    from . import _lib

    rng = mask_generator.rng
    version, internal, gauss = rng.getstate()
    state = (C.c_uint32 * 624)(*internal[:624])
    pos = C.c_int(internal[624])
    edges = np.ascontiguousarray(np.linspace(mask_ratio_min, mask_ratio_max, n_masked_crops + 1), dtype=np.float64)
    H, W = mask_generator.get_shape()
    out = np.zeros((n_crops, H, W), dtype=np.uint8)
    rc = _lib.load().lt_sample_block_masks(C.cast(state, C.c_void_p), C.cast(C.pointer(pos), C.c_void_p), edges.ctypes.data, n_masked_crops, n_crops, H, W,
                                           mask_generator.max_num_patches, mask_generator.min_num_patches, mask_generator.log_aspect_ratio[0],
                                           mask_generator.log_aspect_ratio[1], out.ctypes.data)
    _lib.check(rc, "lt_sample_block_masks")
    rng.setstate((version, tuple(state) + (pos.value,), gauss))
    return out.astype(bool)


def producer_main(key: Tuple[Any, ...], rng_state: Any, conn: Any) -> None:
    """Body of the producer process: sample step after step and push the grids down the pipe (a full pipe blocks the send, so
    the process stays one or two steps ahead and then sleeps)."""
    mn, mx, n_masked, n_crops, grid = key
    rng = random.Random()
    rng.setstate(rng_state)
    gen = MaskingGenerator(input_size=tuple(grid), max_num_patches=int(0.5 * grid[0] * grid[1]), rng=rng)
    try:
        while True:
            conn.send(sample_mask_grids(mn, mx, n_masked, n_crops, gen))
    except (BrokenPipeError, EOFError, OSError, KeyboardInterrupt):
        pass
