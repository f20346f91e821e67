"""TEST INFRASTRUCTURE ONLY.  CPU restatement (plain torch fp32) of the DistillationV3 training step with a frozen DINOv3 ViT
teacher and a DINOv2-ViT student (reference LT/_methods/distillationv3/distillationv3.py:235-374 training_step_impl,
_mixup_data, _forward_teacher, _forward_student, _update_queue; distillationv3_loss.py:35-117; Method.configure_optimizers
LT/_methods/method.py:89-121; parameter partition LT/_optim/optimizer_helpers.py:56-175; clip 1.0 distillationv3.py:400-410).
Pinned against tests/golden/distill_v3_d64.pt, which oracle/make_golden.py writes by running the reference's own
DistillationV3 class through oracle/ref_harness.py.  Un-vendored: CosineWarmupScheduler (lightly), restated in dinov2_oracle."""
from __future__ import annotations

import math
from typing import Any, Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F
from torch import Tensor

from . import dinov2_oracle as O2
from . import dinov3_oracle as O3
from . import resnet_oracle as OR

NO_DECAY_KEYS = ("cls_token", "mask_token", "storage_token", "register_token", "pos_embed")


def decays(name: str, p: Tensor) -> bool:
    """optimizer_helpers.get_weight_decay_parameters for the modules on this path: norm layers, biases, <= 1-D parameters
    (LayerScale gamma), tokens and positional embeddings are not decayed."""
    if p.ndim <= 1 or "norm" in name or name.endswith("bias") or any(k in name for k in NO_DECAY_KEYS):
        return False
    return True


def distill_loss(tg: Tensor, tl: Tensor, sg: Tensor, sl: Tensor, queue: Tensor, temp_g: float, temp_l: float) -> Tuple[Tensor, Tensor]:
    kl = torch.nn.KLDivLoss(reduction="batchmean", log_target=False)
    g = kl(F.log_softmax(sg @ queue.t() / temp_g, dim=-1), F.softmax(tg @ queue.t() / temp_g, dim=-1))
    tt = torch.einsum("bmd,bnd->bmn", tl, tl).flatten(0, 1)
    ss = torch.einsum("bmd,bnd->bmn", sl, sl).flatten(0, 1)
    l = kl(F.log_softmax(ss / temp_l, dim=-1), F.softmax(tt / temp_l, dim=-1))
    return g, l


class OracleDistillationV3:
    def __init__(self, student_backbone: Dict[str, Tensor], student_cfg: Dict[str, Any], teacher_state: Dict[str, Tensor],
                 teacher_cfg: Dict[str, Any], proj_global: Dict[str, Tensor], proj_local: Dict[str, Tensor], queue_size: int,
                 global_batch_size: int, total_steps: int, max_epochs: int = 1, temperature_global: float = 0.07,
                 temperature_local: float = 0.07, loss_local_weight: float = 1.0, lr: float = 0.0005, weight_decay: float = 0.04,
                 reference_batch_size: int = 1536) -> None:
        self.resnet = None
        if student_cfg.get("kind") == "resnet":
            # convolutional student (restated torchvision ResNet, oracle/resnet_oracle.py) in train() mode: batch-statistics BatchNorm.
            # The trained parameters are those the reference's ResNetModelWrapper registers: conv1 .. layer4 (not the classifier).
            self.resnet = OR.ResNet(tuple(student_cfg["layers"]), width=student_cfg.get("width", 64))
            self.resnet.load_state_dict({k: v.detach().clone() for k, v in student_backbone.items()})
            self.resnet.train()
            self.sb = {n: p_ for n, p_ in self.resnet.named_parameters() if not n.startswith("fc.")}
        else:
            self.sb = {k: (v.detach().clone().requires_grad_(True) if not k.endswith(("bias_mask", "periods")) else v.detach().clone())
                       for k, v in student_backbone.items()}
        self.pg = {k: v.detach().clone().requires_grad_(True) for k, v in proj_global.items()}
        self.pl = {k: v.detach().clone().requires_grad_(True) for k, v in proj_local.items()}
        self.teacher = {k: v.detach().clone() for k, v in teacher_state.items()}
        self.scfg, self.tcfg = student_cfg, teacher_cfg
        self.tg_t, self.tl_t, self.w_local = temperature_global, temperature_local, loss_local_weight
        d_t = teacher_state["cls_token"].shape[-1]
        self.queue = torch.zeros(queue_size, d_t)
        named = [("backbone." + k, v) for k, v in self.sb.items() if v.requires_grad] + [("proj_global." + k, v) for k, v in self.pg.items()] + \
                [("proj_local." + k, v) for k, v in self.pl.items()]
        dec = [p for n, p in named if decays(n, p)]
        nod = [p for n, p in named if not decays(n, p)]
        self.n_decay, self.n_no_decay = len(dec), len(nod)
        scale = math.sqrt(global_batch_size / reference_batch_size)       # lr_scale_method = "sqrt" (method.py:91-93)
        self.opt = torch.optim.AdamW([{"params": dec}, {"params": nod, "weight_decay": 0.0}], lr=lr * scale, betas=(0.9, 0.999),
                                     eps=1e-8, weight_decay=weight_decay)
        warm_epochs = min(10, max_epochs / 10)
        self.warmup = min(int(total_steps), int(total_steps / max_epochs * warm_epochs))
        self.total = int(total_steps)
        self.base_lr = lr * scale
        self.step_idx = 0
        self._set_lr()

    def _set_lr(self) -> None:
        f = O2.cosine_warmup_factor(self.step_idx, self.warmup, self.total, 0.001)   # lightly's default end_value
        for g in self.opt.param_groups:
            g["lr"] = self.base_lr * f

    def forward_loss(self, x: Tensor, lam: float, index: Tensor, rescales: Any = None) -> Tuple[Tensor, Dict[str, float]]:
        x = lam * x + (1.0 - lam) * x[index]
        with torch.no_grad():
            t = O3.dinov3_vit_forward(self.teacher, x, self.tcfg)
            tg = F.normalize(t["x_norm_clstoken"], dim=-1, p=2)
            tl = F.normalize(t["x_norm_patchtokens"], dim=-1, p=2)
        if self.resnet is not None:      # distillationv3.py:324-354 with ResNetModelWrapper.forward_features / forward_pool
            fm = OR.features(self.resnet, x)                                   # [B, C, h, w]
            s = {"cls": self.resnet.avgpool(fm).flatten(1), "patch": fm.permute(0, 2, 3, 1).flatten(1, 2)}
            hw_s = (fm.shape[2], fm.shape[3])
        elif "rope_base" in self.scfg:     # DINOv3 student in training mode: one log-uniform RoPE rescale draw per block, after the
            rmax = math.log(self.scfg["rope_rescale"]) if self.scfg.get("rope_rescale") else None   # mixup draws (same RNG order)
            rs = [torch.empty(1).uniform_(-rmax, rmax).exp() for _ in range(self.scfg["depth"])] if rmax is not None and rescales is None else rescales
            s3 = O3.dinov3_vit_forward(self.sb, x, self.scfg, rescales=rs)
            s = {"cls": s3["x_norm_clstoken"], "patch": s3["x_norm_patchtokens"]}
        else:
            s = O2.vit_forward(self.sb, x, self.scfg, masks=None)
        sg = F.linear(s["cls"], self.pg["weight"], self.pg["bias"])
        sl = F.linear(s["patch"], self.pl["weight"], self.pl["bias"])
        if sl.shape[1] != tl.shape[1]:      # bilinear resize onto the teacher grid (distillationv3.py:338-345)
            ps_t = self.tcfg["patch_size"]
            if self.resnet is not None:
                hs, ws_ = hw_s
            else:
                ps_s = self.scfg["patch_size"]
                hs, ws_ = x.shape[2] // ps_s, x.shape[3] // ps_s
            ht, wt = x.shape[2] // ps_t, x.shape[3] // ps_t
            sl = sl.reshape(sl.shape[0], hs, ws_, -1).permute(0, 3, 1, 2)
            sl = F.interpolate(sl, size=(ht, wt), mode="bilinear", align_corners=False)
            sl = sl.permute(0, 2, 3, 1).flatten(1, 2)
        sg, sl = F.normalize(sg, dim=-1, p=2), F.normalize(sl, dim=-1, p=2)
        B, Q = tg.shape[0], self.queue.shape[0]
        with torch.no_grad():                                   # _update_queue, distillationv3.py:275-291
            if B >= Q:
                self.queue = tg[:Q].clone()
            else:
                self.queue[B:] = self.queue[:-B].clone()
                self.queue[:B] = tg
        g, l = distill_loss(tg, tl, sg, sl, self.queue, self.tg_t, self.tl_t)
        loss = g + self.w_local * l
        return loss, {"global_loss": float(g.detach()), "local_loss": float(l.detach())}

    def train_step(self, x: Tensor, lam: float, index: Tensor, rescales: Any = None) -> Dict[str, float]:
        loss, logs = self.forward_loss(x, lam, index, rescales)
        loss.backward()
        params = [p for g in self.opt.param_groups for p in g["params"]]
        gnorm = torch.nn.utils.clip_grad_norm_(params, 1.0)
        self.opt.step()
        self.opt.zero_grad(set_to_none=True)
        self.step_idx += 1
        self._set_lr()
        logs.update(loss=float(loss.detach()), grad_norm=float(gnorm))
        return logs


class OracleDistillation12:
    """Distillation (v1, LT/_methods/distillation/distillation.py:198-266 + distillation_loss.py:38-75) and DistillationV2
    (LT/_methods/distillationv2/distillationv2.py:196-289 + distillationv2_loss.py:27-44) with a DINOv2 ViT teacher and student,
    AdamW, the generic Method.configure_optimizers schedule, clip 1.0.  Pinned against tests/golden/distill_v{1,2}_d64.pt, written by
    the reference's own classes (oracle/make_golden.py::make_distill12)."""

    def __init__(self, kind: str, student_backbone: Dict[str, Tensor], student_cfg: Dict[str, Any], teacher_state: Dict[str, Tensor],
                 teacher_cfg: Dict[str, Any], head: Dict[str, Tensor], queue_size: int, global_batch_size: int, total_steps: int, max_epochs: int = 1,
                 temperature: float = 0.07, n_teacher_blocks: int = 2, lr: float = 0.0005, weight_decay: float = 0.0,
                 reference_batch_size: int = 1536, optimizer: str = "adamw", lars: Optional[Dict[str, Any]] = None) -> None:
        self.kind = kind
        self.sb = {k: v.detach().clone().requires_grad_(True) for k, v in student_backbone.items()}
        self.head = {k: v.detach().clone().requires_grad_(True) for k, v in head.items()}
        self.teacher = {k: v.detach().clone() for k, v in teacher_state.items()}
        self.scfg, self.tcfg, self.temp, self.nb = student_cfg, teacher_cfg, temperature, n_teacher_blocks
        self.queue = torch.zeros(queue_size, teacher_state["cls_token"].shape[-1])
        named = [("backbone." + k, v) for k, v in self.sb.items()] + [("head." + k, v) for k, v in self.head.items()]
        dec = [p for n, p in named if decays(n, p)]
        nod = [p for n, p in named if not decays(n, p)]
        self.n_decay, self.n_no_decay = len(dec), len(nod)
        scale = math.sqrt(global_batch_size / reference_batch_size)
        groups = [{"params": dec}, {"params": nod, "weight_decay": 0.0}]
        if optimizer == "lars":   # the reference's "auto" optimizer (DistillationLARSArgs, distillation.py:140-147): oracle/lars_oracle.py
            from oracle.lars_oracle import LARS
            kw = dict(momentum=0.9, dampening=0.0, nesterov=False, trust_coefficient=0.001, eps=1e-8)
            kw.update(lars or {})
            self.opt = LARS(groups, lr=lr * scale, weight_decay=weight_decay, **kw)
        else:
            self.opt = torch.optim.AdamW(groups, lr=lr * scale, betas=(0.9, 0.999), eps=1e-8, weight_decay=weight_decay)
        warm_epochs = min(10, max_epochs / 10)
        self.warmup = min(int(total_steps), int(total_steps / max_epochs * warm_epochs))
        self.total, self.base_lr, self.step_idx = int(total_steps), lr * scale, 0
        self._set_lr()

    def _set_lr(self) -> None:
        f = O2.cosine_warmup_factor(self.step_idx, self.warmup, self.total, 0.001)
        for g in self.opt.param_groups:
            g["lr"] = self.base_lr * f

    def forward_loss(self, x: Tensor, lam: float, index: Tensor) -> Tensor:
        x = lam * x + (1.0 - lam) * x[index]
        w = self.head.get("weight", self.head.get("mlp.weight"))
        bias = self.head.get("bias", self.head.get("mlp.bias"))
        if self.kind == "v1":
            with torch.no_grad():
                tg = F.normalize(O2.vit_forward(self.teacher, x, self.tcfg)["cls"], dim=-1, p=2)
            sg = F.normalize(F.linear(O2.vit_forward(self.sb, x, self.scfg)["cls"], w, bias), dim=-1, p=2)
            B, Q = tg.shape[0], self.queue.shape[0]
            with torch.no_grad():
                if B >= Q:
                    self.queue = tg[:Q].clone()
                else:
                    self.queue[B:] = self.queue[:-B].clone()
                    self.queue[:B] = tg
            kl = torch.nn.KLDivLoss(reduction="batchmean", log_target=False)
            return kl(F.log_softmax(sg @ self.queue.t() / self.temp, dim=-1), F.softmax(tg @ self.queue.t() / self.temp, dim=-1))
        with torch.no_grad():
            cap: Dict[str, Tensor] = {}
            O2.vit_forward(self.teacher, x, self.tcfg, capture=cap)
            depth = self.tcfg["depth"]
            feats = [F.layer_norm(cap[f"block{i}"], (cap[f"block{i}"].shape[-1],), self.teacher["norm.weight"], self.teacher["norm.bias"], 1e-6)[:, 1:]
                     for i in range(depth - self.nb, depth)]
            tf = torch.cat(feats, dim=-1)                                           # [B, n_pt, nb * Dt]
        sp = F.linear(O2.vit_forward(self.sb, x, self.scfg)["patch"], w, bias)      # [B, n_ps, nb * Dt]
        ps_s, ps_t = self.scfg["patch_size"], self.tcfg["patch_size"]
        hs, ws_, ht, wt = x.shape[2] // ps_s, x.shape[3] // ps_s, x.shape[2] // ps_t, x.shape[3] // ps_t
        sp = sp.reshape(sp.shape[0], hs, ws_, -1).permute(0, 3, 1, 2)
        sp = F.interpolate(sp, size=(ht, wt), mode="bilinear", align_corners=False).permute(0, 2, 3, 1).flatten(1, 2)
        return F.mse_loss(tf, sp)

    def train_step(self, x: Tensor, lam: float, index: Tensor) -> Dict[str, float]:
        loss = self.forward_loss(x, lam, index)
        loss.backward()
        gnorm = torch.nn.utils.clip_grad_norm_([p for g in self.opt.param_groups for p in g["params"]], 1.0)
        self.opt.step()
        self.opt.zero_grad(set_to_none=True)
        self.step_idx += 1
        self._set_lr()
        return {"loss": float(loss.detach()), "grad_norm": float(gnorm)}
