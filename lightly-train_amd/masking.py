"""Host-side iBOT block masking (mirrors MaskingGenerator / create_collated_masks,
LT/_methods/dinov2/utils.py:41-152).  RNG is Python's `random`, consumed in exactly the reference's
order so that `random.seed(s)` reproduces the reference's masks bit for bit."""
from __future__ import annotations

import multiprocessing
import random
from typing import Any, Dict, Optional, Tuple

import numpy as np
import torch

from ._mask_sampler import MaskingGenerator, producer_main, sample_mask_grids, sample_mask_grids_native

__all__ = ["MaskingGenerator", "MaskProducer", "collate_mask_grids", "create_collated_masks"]


def collate_mask_grids(grids: np.ndarray) -> Dict[str, torch.Tensor]:
    collated = torch.from_numpy(grids).flatten(1)
    indices = collated.flatten().nonzero().flatten()
    per_crop = 1.0 / collated.sum(-1).clamp(min=1.0)
    weights = per_crop.unsqueeze(-1).expand_as(collated)[collated]
    return {"collated_masks": collated, "mask_indices_list": indices, "masks_weight": weights}


def create_collated_masks(mask_ratio_min: float, mask_ratio_max: float, n_masked_crops: int, n_crops: int,
                          mask_generator: MaskingGenerator, native: bool = True) -> Dict[str, torch.Tensor]:
    """native=True: the C++ sampler of liblt_amd.so on the same `random` stream (bit-identical masks and stream position, tested);
    native=False: the Python loop of the reference."""
    sample = sample_mask_grids_native if native else sample_mask_grids
    return collate_mask_grids(sample(mask_ratio_min, mask_ratio_max, n_masked_crops, n_crops, mask_generator))


class MaskProducer:
    """Samples the masks of the coming steps in a background PROCESS (the reference samples them in pure
    Python on the training thread at the top of every step, `dinov2.py:300-312` -> `utils.py:41-152`, 13 ms for 256 crops).

    The child owns a private `random.Random` that continues the global `random` stream from the state it had when the
    producer was created, so step k gets exactly the masks the k-th in-line call would have sampled (the global stream itself
    is no longer consumed by mask sampling afterwards).  A process, not a thread: the sampler is pure Python, and a thread
    holding the GIL for 13 ms per step starves the training thread's ~1000 short launch calls (measured: 145 instead of 101
    ms/step).  The child is spawned (no fork of a process that holds a HIP context), imports only the torch-free sampler and is
    throttled by the pipe's buffer."""

    def __init__(self, mask_ratio_min: float, mask_ratio_max: float, n_masked_crops: int, n_crops: int, grid: Tuple[int, int],
                 rng_state: Optional[Any] = None) -> None:
        self.key = (mask_ratio_min, mask_ratio_max, n_masked_crops, n_crops, tuple(grid))
        ctx = multiprocessing.get_context("spawn")
        self._recv, send = ctx.Pipe(duplex=False)
        self._proc = ctx.Process(target=producer_main, name="lt-mask-producer", daemon=True,
                                 args=(self.key, rng_state if rng_state is not None else random.getstate(), send))
        self._proc.start()
        send.close()   # the child holds the only write end: its death shows up as EOF here

    def get(self) -> Dict[str, torch.Tensor]:
        try:
            grids = self._recv.recv()
        except EOFError as e:
            raise RuntimeError(f"mask producer process died (exit code {self._proc.exitcode})") from e
        return collate_mask_grids(grids)

    def close(self) -> None:
        self._recv.close()
        if self._proc.is_alive():
            self._proc.terminate()
        self._proc.join(timeout=5.0)
