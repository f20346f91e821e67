"""Flat fp32 parameter storage for the fused multi-tensor kernels (AdamW, EMA, grad-norm, all-reduce).

Every tensor lives in one contiguous fp32 buffer, padded to 1024-element chunks (a chunk never straddles
two tensors), next to a same-shaped grad buffer and a bf16 shadow (the MFMA operand copy).  The named views
keep the reference's state_dict keys, so checkpoints/exports stay compatible."""
from __future__ import annotations

from typing import Dict, Iterable, List, Tuple

import torch
from torch import Tensor

CHUNK = 1024


class FlatParams:
    def __init__(self, named: Iterable[Tuple[str, Tensor]], device: torch.device | str, with_grad: bool = True) -> None:
        named = list(named)
        self.names: List[str] = [n for n, _ in named]
        self.shapes: Dict[str, torch.Size] = {n: t.shape for n, t in named}
        self.offsets: Dict[str, int] = {}
        off = 0
        seg_of_chunk: List[int] = []
        for i, (n, t) in enumerate(named):
            self.offsets[n] = off
            nchunks = (t.numel() + CHUNK - 1) // CHUNK
            seg_of_chunk += [i] * nchunks
            off += nchunks * CHUNK
        self.numel = off
        self.device = torch.device(device)
        self.data = torch.zeros(off, dtype=torch.float32, device=self.device)
        self.bf16 = torch.zeros(off, dtype=torch.bfloat16, device=self.device)
        self.grad = torch.zeros(off, dtype=torch.float32, device=self.device) if with_grad else None
        self.seg_of_chunk = torch.tensor(seg_of_chunk, dtype=torch.int32, device=self.device)
        self.p: Dict[str, Tensor] = {}
        self.b: Dict[str, Tensor] = {}
        self.g: Dict[str, Tensor] = {}
        for n, t in named:
            o, k = self.offsets[n], t.numel()
            self.p[n] = self.data[o:o + k].view(t.shape)
            self.b[n] = self.bf16[o:o + k].view(t.shape)
            if with_grad:
                self.g[n] = self.grad[o:o + k].view(t.shape)
            self.p[n].copy_(t.to(device=self.device, dtype=torch.float32))
        self.bf16.copy_(self.data)  # one-time init cast (torch plumbing); steady-state casts are fused into AdamW/EMA

    def span(self, prefixes: Tuple[str, ...]) -> Tuple[int, int]:
        """[lo, hi) of the flat buffers that holds exactly the tensors whose names start with one of `prefixes`
        (ValueError when those tensors are not contiguous in the flat order)."""
        idx = [i for i, n in enumerate(self.names) if n.startswith(prefixes)]
        if not idx or idx != list(range(idx[0], idx[-1] + 1)):
            raise ValueError(f"parameters {prefixes} are not one contiguous run of the flat buffer")
        hi = self.offsets[self.names[idx[-1] + 1]] if idx[-1] + 1 < len(self.names) else self.numel
        return self.offsets[self.names[idx[0]]], hi

    def state_dict(self, prefix: str = "") -> Dict[str, Tensor]:
        return {prefix + n: self.p[n].detach().clone() for n in self.names}

    def load_state_dict(self, sd: Dict[str, Tensor], prefix: str = "") -> None:
        for n in self.names:
            self.p[n].copy_(sd[prefix + n].to(self.device, torch.float32))
        self.bf16.copy_(self.data)
