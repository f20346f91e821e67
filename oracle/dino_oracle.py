"""TEST INFRASTRUCTURE (only tests/, smoke() and bench.py's cpu_baseline leg may import this).

The LightlySSL pieces the reference's DINO method imports (LT/_methods/dino/dino.py:15-17):
    from lightly.loss import DINOLoss
    from lightly.models.modules.heads import DINOProjectionHead
    from lightly.models.utils import get_weight_decay_parameters

PARITY UNPINNED for these three: `lightly` (pyproject.toml:32 `lightly>=1.5.26`) is not vendored under /root/reference and not installed
here.  They are restated from the published package and anchored on what the reference tree itself holds:
  * the call sites in dino.py (constructor arguments :239-269, `criterion(teacher_out=.., student_out=.., teacher_temp=..)` :311-316,
    `student_projection_head.last_layer.parameters()` :361, the group names :373-388);
  * the reference's VENDORED twin of the loss, `_methods/dinov2/dinov2_loss.py:61-145` (softmax((t - center) / T_t), the
    cross-entropy against log_softmax(s / T_s), center <- m center + (1 - m) mean(t)): tests/test_oracle_pin.py checks that DINOLoss
    below equals that class for two global views, and `oracle/make_golden.py:make_dino_v1` asserts it while writing the fixture;
  * the reference's VENDORED twin of the head, `_methods/dinov2/dinov2_head.py:32-71` (3-layer GELU MLP, trunc-normal 0.02 / zero bias,
    L2-normalised bottleneck, weight-normed prototype layer with g = 1) -- same function, different attribute names.

ref_harness.install() registers them at the import paths above, so that the reference's own `DINO` class runs end to end and writes
tests/golden/dino_v1_d64*.pt.
"""
from __future__ import annotations

from typing import Iterable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist
import torch.nn.functional as F
from torch import Tensor, nn
from torch.nn.modules.batchnorm import _NormBase

_NORM_LAYERS = (_NormBase, nn.LayerNorm, nn.CrossMapLRN2d, nn.LocalResponseNorm, nn.GroupNorm)


def get_weight_decay_parameters(modules: Iterable[nn.Module], decay_norm: bool = False, decay_bias: bool = False,
                                norm_layers: Tuple[type, ...] = _NORM_LAYERS) -> Tuple[List[nn.Parameter], List[nn.Parameter]]:
    """lightly.models.utils.get_weight_decay_parameters: (decayed, not decayed).  Not decayed: every parameter of a normalisation layer and
    every parameter whose own name contains "bias".  Everything else -- tokens, positional embeddings, LayerScale -- IS decayed (the
    reference's newer `optimizer_helpers.get_weight_decay_parameters`, optimizer_helpers.py:83-150, which also exempts those, is not the
    one dino.py imports)."""
    params: List[nn.Parameter] = []
    params_no_weight_decay: List[nn.Parameter] = []
    for module in modules:
        for mod in module.modules():
            if isinstance(mod, norm_layers):
                (params if decay_norm else params_no_weight_decay).extend(mod.parameters(recurse=False))
            else:
                for name, param in mod.named_parameters(recurse=False):
                    if "bias" in name:
                        (params if decay_bias else params_no_weight_decay).append(param)
                    else:
                        params.append(param)
    return params, params_no_weight_decay


class DINOProjectionHead(nn.Module):
    """lightly.models.modules.heads.DINOProjectionHead: `layers` = Linear, GELU, Linear, GELU, Linear (Sequential indices 0, 2, 4),
    F.normalize, `last_layer` = weight_norm(Linear(bottleneck, output, bias=False)) with weight_g filled with 1 and, for
    norm_last_layer=True, excluded from training.  batch_norm=True (ONE BatchNorm1d instance placed after both hidden Linear layers,
    which then have no bias) is not restated."""

    def __init__(self, input_dim: int = 2048, hidden_dim: int = 2048, bottleneck_dim: int = 256, output_dim: int = 65536, batch_norm: bool = False,
                 freeze_last_layer: int = -1, norm_last_layer: bool = True) -> None:
        super().__init__()
        if batch_norm:
            raise NotImplementedError("DINOProjectionHead(batch_norm=True) is not restated")
        self.layers = nn.Sequential(nn.Linear(input_dim, hidden_dim), nn.GELU(), nn.Linear(hidden_dim, hidden_dim), nn.GELU(),
                                    nn.Linear(hidden_dim, bottleneck_dim))
        for m in self.layers:
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02)
                nn.init.constant_(m.bias, 0)
        self.freeze_last_layer = freeze_last_layer
        self.last_layer = nn.utils.weight_norm(nn.Linear(bottleneck_dim, output_dim, bias=False))
        self.last_layer.weight_g.data.fill_(1)
        if norm_last_layer:
            self.last_layer.weight_g.requires_grad = False

    def cancel_last_layer_gradients(self, current_epoch: int) -> None:
        if current_epoch >= self.freeze_last_layer:
            return
        for param in self.last_layer.parameters():
            param.grad = None

    def forward(self, x: Tensor) -> Tensor:
        x = self.layers(x)
        x = F.normalize(x, dim=-1, p=2)
        return self.last_layer(x)


class Center(nn.Module):
    """lightly.models.modules.center.Center (mode "mean"): buffer `center`, moved towards the mean of the teacher outputs over the view
    and batch dimensions (averaged over ranks) with the given momentum."""

    def __init__(self, size: Sequence[int], mode: str = "mean", momentum: float = 0.9) -> None:
        super().__init__()
        if mode != "mean":
            raise ValueError(f"Invalid center mode: {mode}")
        self.register_buffer("center", torch.zeros(tuple(size)))
        self.dim = tuple(i for i, s in enumerate(size) if s == 1)
        self.momentum = momentum

    @property
    def value(self) -> Tensor:
        return self.center

    @torch.no_grad()
    def update(self, x: Tensor) -> None:
        batch_center = torch.mean(x, dim=self.dim, keepdim=True)
        if dist.is_available() and dist.is_initialized():
            dist.all_reduce(batch_center)
            batch_center = batch_center / dist.get_world_size()
        self.center = self.center * self.momentum + batch_center * (1 - self.momentum)


class DINOLoss(nn.Module):
    """lightly.loss.DINOLoss.  forward(teacher_out = T tensors [B, K], student_out = S tensors [B, K], teacher_temp):
        t = softmax((stack(teacher_out) - center) / teacher_temp);  s = log_softmax(stack(student_out) / student_temp)
        loss[t, s] = -sum_{b,k} t[t, b, k] s[s, b, k], diagonal zeroed;  result = sum(loss) / ((T S - min(T, S)) B)
    then the center update from the raw teacher outputs."""

    def __init__(self, output_dim: int = 65536, warmup_teacher_temp: float = 0.04, teacher_temp: float = 0.04, warmup_teacher_temp_epochs: int = 30,
                 student_temp: float = 0.1, center_momentum: float = 0.9, center_mode: str = "mean") -> None:
        super().__init__()
        self.teacher_temp = teacher_temp
        self.student_temp = student_temp
        self.warmup_teacher_temp_epochs = warmup_teacher_temp_epochs
        self.teacher_temp_schedule = torch.linspace(warmup_teacher_temp, teacher_temp, warmup_teacher_temp_epochs)
        self.center = Center(size=(1, 1, output_dim), mode=center_mode, momentum=center_momentum)

    def forward(self, teacher_out: Sequence[Tensor], student_out: Sequence[Tensor], teacher_temp: Optional[float] = None,
                epoch: Optional[int] = None) -> Tensor:
        if teacher_temp is None:
            if epoch is None:
                raise ValueError("teacher_temp or epoch must be given")
            teacher_temp = float(self.teacher_temp_schedule[epoch]) if epoch < self.warmup_teacher_temp_epochs else self.teacher_temp
        t_stack = torch.stack(list(teacher_out))
        t_out = F.softmax((t_stack - self.center.value) / teacher_temp, dim=-1)
        s_out = F.log_softmax(torch.stack(list(student_out)) / self.student_temp, dim=-1)
        loss = -torch.einsum("tbd,sbd->ts", t_out, s_out)
        loss.fill_diagonal_(0)
        n_terms = loss.numel() - loss.diagonal().numel()
        loss = loss.sum() / (n_terms * teacher_out[0].shape[0])
        self.center.update(t_stack)
        return loss
