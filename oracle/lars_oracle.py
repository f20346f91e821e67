"""TEST INFRASTRUCTURE (only tests/, smoke() and bench.py's cpu_baseline leg may import this).

LARS as the reference's `LARSArgs.get_optimizer` builds it (LT/_optim/lars_args.py:12,21-37: `from lightly.utils.lars import LARS`,
kwargs lr, momentum, dampening, weight_decay, nesterov, trust_coefficient, eps) -- the "auto" optimizer of Distillation and
DistillationV2 (distillation.py:140-147,294; distillationv2.py:106,310) and an option of DistillationV3 (distillationv3.py:147-157,386).

PARITY UNPINNED: `lightly.utils.lars` is LightlySSL code (dependency `lightly`, not vendored under /root/reference and not installed
here).  Restated from its published algorithm: layer-wise adaptive rate scaling (You, Gitman, Ginsburg 2017) in the form LightlySSL
took from PyTorch Lightning Bolts -- per parameter tensor, and only for groups with weight_decay != 0 and non-zero norms,
    lars_lr = trust_coefficient * ||p|| / (||g|| + weight_decay * ||p|| + eps);   d = (g + weight_decay * p) * lars_lr
followed by torch.optim.SGD's momentum rule (buffer initialised with the first d; buf = momentum * buf + (1 - dampening) * d;
nesterov: d + momentum * buf) and p -= lr * d.  Groups with weight_decay == 0 (biases, norm layers: optimizer_helpers.py:56-77)
therefore take plain momentum-SGD steps.  ref_harness.install() registers this class as `lightly.utils.lars.LARS`, so that the
reference's own Distillation class runs with its "auto" optimizer arguments to write tests/golden/distill_v1_d64_lars.pt.
"""
from __future__ import annotations

import torch
from torch.optim.optimizer import Optimizer


class LARS(Optimizer):
    def __init__(self, params, lr: float, momentum: float = 0.0, dampening: float = 0.0, weight_decay: float = 0.0, nesterov: bool = False,
                 trust_coefficient: float = 0.001, eps: float = 1e-8) -> None:
        if nesterov and (momentum <= 0 or dampening != 0):
            raise ValueError("Nesterov momentum requires a momentum and zero dampening")
        super().__init__(params, dict(lr=lr, momentum=momentum, dampening=dampening, weight_decay=weight_decay, nesterov=nesterov,
                                      trust_coefficient=trust_coefficient, eps=eps))

    @torch.no_grad()
    def step(self, closure=None):  # type: ignore[no-untyped-def]
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            wd, mom, damp, nesterov = group["weight_decay"], group["momentum"], group["dampening"], group["nesterov"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                d_p = p.grad
                p_norm, g_norm = torch.norm(p.data), torch.norm(p.grad.data)
                if wd != 0 and p_norm != 0 and g_norm != 0:
                    lars_lr = p_norm / (g_norm + p_norm * wd + group["eps"]) * group["trust_coefficient"]
                    d_p = d_p.add(p, alpha=wd) * lars_lr
                if mom != 0:
                    st = self.state[p]
                    if "momentum_buffer" not in st:
                        buf = st["momentum_buffer"] = torch.clone(d_p).detach()
                    else:
                        buf = st["momentum_buffer"]
                        buf.mul_(mom).add_(d_p, alpha=1 - damp)
                    d_p = d_p.add(buf, alpha=mom) if nesterov else buf
                p.add_(d_p, alpha=-group["lr"])
        return loss
