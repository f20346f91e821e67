"""Host-side iBOT block masking (mirrors MaskingGenerator / create_collated_masks,
LT/_methods/dinov2/utils.py:41-152).  RNG is Python's `random`, consumed in exactly the reference's
order so that `random.seed(s)` reproduces the reference's masks bit for bit."""
from __future__ import annotations

import math
import random
from typing import Dict, List, Tuple

import numpy as np
import torch


class MaskingGenerator:
    def __init__(self, input_size: int | Tuple[int, int], max_num_patches: int, min_num_patches: int = 4,
                 min_aspect: float = 0.3, max_aspect: float | None = None) -> None:
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
            area = random.uniform(self.min_num_patches, budget)
            aspect = math.exp(random.uniform(*self.log_aspect_ratio))
            bh = int(round(math.sqrt(area * aspect)))
            bw = int(round(math.sqrt(area / aspect)))
            if bw < self.width and bh < self.height:
                y0 = random.randint(0, self.height - bh)
                x0 = random.randint(0, self.width - bw)
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
        target = int(n_tokens * random.uniform(edges[i], edges[i + 1]))
        grids.append(torch.from_numpy(mask_generator(target)))
    for _ in range(n_masked_crops, n_crops):
        grids.append(torch.from_numpy(mask_generator(0)))
    random.shuffle(grids)
    collated = torch.stack(grids).flatten(1)
    indices = collated.flatten().nonzero().flatten()
    per_crop = 1.0 / collated.sum(-1).clamp(min=1.0)
    weights = per_crop.unsqueeze(-1).expand_as(collated)[collated]
    return {"collated_masks": collated, "mask_indices_list": indices, "masks_weight": weights}
