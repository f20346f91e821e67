"""Host-side iBOT block masking (mirrors MaskingGenerator / create_collated_masks,
LT/_methods/dinov2/utils.py:41-152).  RNG is Python's `random`, consumed in exactly the reference's
order so that `random.seed(s)` reproduces the reference's masks bit for bit."""
from __future__ import annotations

import math
import queue
import random
import threading
from typing import Any, Dict, List, Optional, Tuple

import numpy as np
import torch


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


def create_collated_masks(mask_ratio_min: float, mask_ratio_max: float, n_masked_crops: int, n_crops: int,
                          mask_generator: MaskingGenerator) -> Dict[str, torch.Tensor]:
    n_tokens = mask_generator.num_patches
    edges = np.linspace(mask_ratio_min, mask_ratio_max, n_masked_crops + 1)
    grids: List[torch.Tensor] = []
    for i in range(n_masked_crops):
        target = int(n_tokens * mask_generator.rng.uniform(edges[i], edges[i + 1]))
        grids.append(torch.from_numpy(mask_generator(target)))
    for _ in range(n_masked_crops, n_crops):
        grids.append(torch.from_numpy(mask_generator(0)))
    mask_generator.rng.shuffle(grids)
    collated = torch.stack(grids).flatten(1)
    indices = collated.flatten().nonzero().flatten()
    per_crop = 1.0 / collated.sum(-1).clamp(min=1.0)
    weights = per_crop.unsqueeze(-1).expand_as(collated)[collated]
    return {"collated_masks": collated, "mask_indices_list": indices, "masks_weight": weights}


class MaskProducer:
    """Samples the masks of the coming steps on a background thread (the reference samples them in pure
    Python on the training thread at the top of every step, `dinov2.py:300-312` -> `utils.py:41-152`).

    The thread owns a private `random.Random` that continues the global `random` stream from the state it had when the
    producer was created, so step k gets exactly the masks the k-th in-line call would have sampled (the global stream itself
    is no longer consumed by mask sampling afterwards).  The sampling is pure Python (GIL-bound), but the training thread
    spends its time inside ctypes / torch calls that release the GIL, so the two overlap."""

    def __init__(self, mask_ratio_min: float, mask_ratio_max: float, n_masked_crops: int, n_crops: int, grid: Tuple[int, int],
                 depth: int = 2, rng_state: Optional[Any] = None) -> None:
        self.key = (mask_ratio_min, mask_ratio_max, n_masked_crops, n_crops, tuple(grid))
        self._rng = random.Random()
        self._rng.setstate(rng_state if rng_state is not None else random.getstate())
        self._gen = MaskingGenerator(input_size=tuple(grid), max_num_patches=int(0.5 * grid[0] * grid[1]), rng=self._rng)
        self._q: "queue.Queue[Any]" = queue.Queue(maxsize=depth)
        self._stop = threading.Event()
        self._thread = threading.Thread(target=self._run, name="lt-mask-producer", daemon=True)
        self._thread.start()

    def _run(self) -> None:
        mn, mx, n_masked, n_crops, _ = self.key
        while not self._stop.is_set():
            try:
                item: Any = create_collated_masks(mn, mx, n_masked, n_crops, self._gen)
            except BaseException as e:   # surface the failure on the consumer side instead of dying silently
                item = e
            while not self._stop.is_set():
                try:
                    self._q.put(item, timeout=0.1)
                    break
                except queue.Full:
                    continue
            if isinstance(item, BaseException):
                return

    def get(self) -> Dict[str, torch.Tensor]:
        item = self._q.get()
        if isinstance(item, BaseException):
            raise item
        return item

    def close(self) -> None:
        self._stop.set()
        self._thread.join(timeout=5.0)
