"""LARS on flat parameter storage: the reference's `LARSArgs` (LT/_optim/lars_args.py:21-37) around `lightly.utils.lars.LARS`, the "auto"
optimizer of Distillation and DistillationV2 (distillation.py:140-147,294, distillationv2.py:106,310) and an option of DistillationV3
(distillationv3.py:147-157,386).  The rule itself (csrc/optim.hip: lt_lars_norms + lt_lars_flat; stated in include/lt_amd.h): parameter
tensors of the no-weight-decay group (biases, norm layers, tokens: optimizer_helpers.py:56-77) take plain momentum-SGD steps, as the
optimizer's `weight_decay != 0` gate implies."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch
from torch import Tensor

from . import ops
from .params import FlatParams


@dataclass
class LARSArgs:
    """DistillationLARSArgs / DistillationV2LARSArgs / DistillationV3LARSArgs: identical values in all three methods."""
    lr: float = 1.8  # 0.3 * 1536 / 256
    momentum: float = 0.9
    dampening: float = 0.0
    weight_decay: float = 1e-6
    nesterov: bool = False
    trust_coefficient: float = 0.001
    eps: float = 1e-8


class FlatLARS:
    def __init__(self, fp: FlatParams, args: LARSArgs) -> None:
        if args.nesterov and (args.momentum <= 0 or args.dampening != 0):
            raise ValueError("Nesterov momentum requires a momentum and zero dampening")
        self.fp, self.args = fp, args
        dev = fp.device
        self.momentum_buffer = torch.zeros_like(fp.data) if args.momentum != 0 else None
        self.ws = torch.empty(2 * fp.numel // 1024, dtype=torch.float32, device=dev)
        self.seg_norms = torch.zeros(len(fp.names), 2, dtype=torch.float32, device=dev)
        begins = [fp.offsets[n] // 1024 for n in fp.names] + [fp.numel // 1024]
        self.seg_chunk_begin = torch.tensor(begins, dtype=torch.int32, device=dev)
        self.steps = 0

    def step(self, seg_lr: Tensor, seg_wd_on: Tensor, lr_factor: float, sumsq: Optional[Tensor], max_norm: float) -> None:
        a, fp = self.args, self.fp
        ops.lars_flat(fp.data, fp.grad, self.momentum_buffer, fp.bf16, fp.seg_of_chunk, self.seg_chunk_begin, seg_lr, seg_wd_on, self.ws, self.seg_norms,
                      lr_factor, a.weight_decay, a.momentum, a.dampening, a.nesterov, a.trust_coefficient, a.eps, self.steps == 0, sumsq, max_norm)
        self.steps += 1

    def state(self) -> dict:
        return dict(momentum_buffer=None if self.momentum_buffer is None else self.momentum_buffer.detach().clone(), steps=self.steps)

    def load_state(self, st: dict) -> None:
        if self.momentum_buffer is not None and st.get("momentum_buffer") is not None:
            self.momentum_buffer.copy_(st["momentum_buffer"].to(self.momentum_buffer.device))
        self.steps = int(st["steps"])
