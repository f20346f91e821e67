"""DINO (v1) on the MI355X kernels -- mirrors lightly_train's `DINO(Method)` (LT/_methods/dino/dino.py:221-480) for ViT backbones.

Same kernels and the same explicit backward as the DINOv2 step (this class derives from `dinov2.DINOv2` for the flat parameter
storage, the ViT / projection-head engines, the 5-stream schedule and the data-parallel gradient reduction); what differs is what the
reference's DINO does differently:

  * the EMA teacher update runs FIRST, at the step's own global_step (dino.py:273-284), for backbone and head;
  * one head per role (`student_projection_head` / `teacher_projection_head`, lightly's DINOProjectionHead: state_dict names
    `layers.{0,2,4}`, `last_layer.weight_g / weight_v`; `norm_last_layer=True` keeps weight_g out of training), fed with the pooled
    class token of every view (`EmbeddingModel(x)` = `forward_pool(forward_features(x))`, dinov2_vit.py:82-103) -- no masking;
  * lightly's DINOLoss: every (teacher view t, student view s != t) pair, / ((T S - min(T, S)) B); the center moves right after the
    loss, from this step's teacher outputs (averaged over ranks);
  * teacher temperature by `_teacher_temp_schedule` (dino.py:485-507), weight decay by a cosine schedule on the group "params" and, once
    it is unfrozen, on the last layer; while `global_step < student_freeze_last_layer_steps` the last layer runs at lr 0 / weight decay 0
    (its momentum buffer / Adam moments still move, as in torch);
  * the optimizer: "auto" = SGD (DINOSGDArgs: lr 0.03, momentum 0.9, weight decay 1e-4) on `lt_sgd_flat`, or AdamW (DINOAdamWArgs:
    lr 5e-4, weight decay 0.04) on `lt_adamw_flat`; decayed = everything that is not a normalisation layer's parameter or a bias
    (lightly's `get_weight_decay_parameters`: tokens, positional embedding and LayerScale ARE decayed here, unlike DINOv2);
    linear lr scaling from batch 256; gradient clipping at 3.0; CosineWarmupScheduler with warmup min(12500, 10 % of the steps).

The LightlySSL pieces are un-vendored in the reference tree (parity unpinned for those: DESIGN.md lists how the test-side restatement
is anchored).  `DINOResNet` below runs the same method on a torchvision ResNet; `batch_norm=True` heads are not built.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Any, Dict, List, Mapping, Optional, Tuple

import torch
import torch.distributed as dist
from torch import Tensor

from . import checkpoint, ops
from .dinov2 import WN_G, WN_V, DINOv2, DINOv2Args, TrainingStepResult
from .schedules import cosine_schedule, warmup_cosine_lr_factor
from .vit import ViTConfig, make_drop_plan

IMAGENET_SIZE = 1_000_000   # LT/_scaling.py:13


def _interpolate(x: float, x0: float, x1: float, v0: float, v1: float, ndigits: int) -> float:
    """LT/_scaling.py:24-41."""
    return round(min(max(v0 + (v1 - v0) * (x - x0) / (x1 - x0), v0), v1), ndigits)


@dataclass
class DINOArgs:
    """DINOArgs (dino.py:47-207) with its "auto" entries at their large-dataset values, + the optimizer arguments (DINOSGDArgs /
    DINOAdamWArgs, dino.py:210-217).  `for_dataset_size` applies the reference's dataset-size scaling (`resolve_auto`)."""
    hidden_dim: int = 2048
    bottleneck_dim: int = 256
    output_dim: int = 65536
    student_freeze_last_layer_steps: int = 1250
    batch_norm: bool = False
    norm_last_layer: bool = True
    teacher_temp: float = 0.07
    warmup_teacher_temp: float = 0.04
    warmup_teacher_temp_steps: int = 37500
    student_temp: float = 0.1
    center_momentum: float = 0.9
    momentum_start: float = 0.996
    momentum_end: float = 1.0
    weight_decay_start: Optional[float] = None    # None = "auto": the optimizer's weight decay (dino.py:201-206)
    weight_decay_end: Optional[float] = None
    warmup_steps: int = 12500
    warmup_max_steps_fraction: float = 0.1
    lr_scale_method: str = "linear"               # MethodArgs defaults (method_args.py:28-29)
    reference_batch_size: int = 256
    gradient_clip_val: float = 3.0                # fixed in configure_gradient_clipping (dino.py:330-340)
    # optimizer
    optimizer: str = "auto"                       # "auto" = "sgd" (dino.py:343-352)
    lr: Optional[float] = None                    # None: 0.03 (SGD) / 0.0005 (AdamW)
    weight_decay: Optional[float] = None          # None: 1e-4 (SGD) / 0.04 (AdamW)
    momentum: float = 0.9                         # SGDArgs (sgd_args.py:19-22)
    betas: Tuple[float, float] = (0.9, 0.999)
    eps: float = 1e-8

    @classmethod
    def for_dataset_size(cls, dataset_size: int, **kw: Any) -> "DINOArgs":
        """`DINOArgs.resolve_auto` (dino.py:80-196): output_dim by bucket, teacher temperatures and the start momentum interpolated
        between a 20k-image dataset and ImageNet."""
        buckets = ((20_000, 1024), (50_000, 2048), (100_000, 4096), (200_000, 16384), (500_000, 32768), (float("inf"), 65536))
        out_dim = next(v for t, v in buckets if dataset_size < t)
        tt = _interpolate(dataset_size, 20_000, IMAGENET_SIZE, 0.02, 0.07, 2)
        wt = min(tt, _interpolate(tt, 0.02, 0.07, 0.02, 0.04, 2))
        m0 = _interpolate(dataset_size, 20_000, IMAGENET_SIZE, 0.99, 0.996, 3)
        d = dict(output_dim=out_dim, teacher_temp=tt, warmup_teacher_temp=wt, momentum_start=m0)
        d.update(kw)
        return cls(**d)


def teacher_temp_schedule(temp: float, warmup_temp: float, warmup_steps: int, step: int) -> float:
    """`_teacher_temp_schedule` with `warmup_steps` given (dino.py:485-507; the epoch-based spelling is deprecated there)."""
    if step < warmup_steps:
        return warmup_temp + step * (temp - warmup_temp) / warmup_steps
    return temp


def decays(flat_name: str) -> bool:
    """lightly.models.utils.get_weight_decay_parameters on the names of a ViT + DINOProjectionHead: parameters of LayerNorm modules
    (`norm1`, `norm2`, `norm`) and parameters named bias are not decayed."""
    parts = flat_name.split(".")
    return not ("bias" in parts[-1] or any(p.startswith("norm") for p in parts[:-1]))


_HEAD_TO_REF = {"mlp.0.": "layers.0.", "mlp.2.": "layers.2.", "mlp.4.": "layers.4.", WN_G: "last_layer.weight_g", WN_V: "last_layer.weight_v"}


def head_key_to_ref(k: str) -> str:
    for a, b in _HEAD_TO_REF.items():
        if k.startswith(a):
            return b + k[len(a):]
    raise KeyError(k)


def head_state_from_ref(sd: Mapping[str, Tensor]) -> Dict[str, Tensor]:
    """A DINOProjectionHead state_dict -> the head engine's names."""
    inv = {v: k for k, v in _HEAD_TO_REF.items()}
    out = {}
    for k, v in sd.items():
        for a, b in inv.items():
            if k.startswith(a):
                out[b + k[len(a):]] = v
                break
        else:
            raise KeyError(f"unexpected projection-head key {k!r}")
    return out


class DINO(DINOv2):
    """The method object; attribute / state_dict layout follows the reference's DINO."""

    supports_accumulation = False   # this step zeroes its gradients and applies the update once per batch

    def __init__(self, vit_cfg: ViTConfig, method_args: Optional[DINOArgs] = None, global_batch_size: int = 256, total_steps: int = 125_000,
                 device: str | torch.device = "cuda", backbone_state: Optional[Dict[str, Tensor]] = None,
                 student_head_state: Optional[Dict[str, Tensor]] = None, teacher_head_state: Optional[Dict[str, Tensor]] = None,
                 teacher_backbone_state: Optional[Dict[str, Tensor]] = None, seed: int = 0) -> None:
        a = method_args or DINOArgs()
        if a.batch_norm:
            raise NotImplementedError("DINO(batch_norm=True): lightly's shared-BatchNorm1d head is not built")
        if a.optimizer not in ("auto", "sgd", "adamw"):
            raise ValueError(f"Invalid optimizer type: '{a.optimizer}'")
        shim = DINOv2Args(hidden_dim=a.hidden_dim, dino_bottleneck_dim=a.bottleneck_dim, output_dim=a.output_dim, batch_norm=False,
                          center_momentum=a.center_momentum, ibot_separate_head=False, lr_scale_method=a.lr_scale_method,
                          reference_batch_size=a.reference_batch_size)
        conv = (lambda s: None if s is None else head_state_from_ref(s) if any(k.startswith("layers.") for k in s) else s)
        super().__init__(vit_cfg, shim, global_batch_size=global_batch_size, total_steps=total_steps, device=device, backbone_state=backbone_state,
                         student_head_state=conv(student_head_state), teacher_head_state=conv(teacher_head_state),
                         teacher_backbone_state=teacher_backbone_state, seed=seed)
        self.method_args = a     # type: ignore[assignment]
        # parameters that never receive a gradient in the reference: the mask token (no masking here) and, with norm_last_layer, weight_g
        self._setup_optimizer(global_batch_size, total_steps, untrained={"backbone.mask_token"} | ({"head." + WN_G} if a.norm_last_layer else set()),
                              decayed=decays)
        self.center = self.dino_center            # [1, K]; `criterion.center.center` [1, 1, K] in the state_dict
        self._gwn = self.student.g["head." + WN_G]

    def _setup_optimizer(self, global_batch_size: int, total_steps: int, untrained: set, decayed: Any) -> None:
        """Learning rate, schedules and the per-tensor group tables of the reference's optimizer (dino.py:343-413): groups `params`
        (decayed), `params_last_layer`, `params_no_weight_decay`; `decayed(flat_name)` says which tensors lightly's
        get_weight_decay_parameters decays for this backbone."""
        a = self.method_args
        self.optimizer = "sgd" if a.optimizer == "auto" else a.optimizer
        lr = a.lr if a.lr is not None else (0.03 if self.optimizer == "sgd" else 0.0005)
        self.weight_decay = a.weight_decay if a.weight_decay is not None else (1e-4 if self.optimizer == "sgd" else 0.04)
        self.wd_start = a.weight_decay_start if a.weight_decay_start is not None else self.weight_decay
        self.wd_end = a.weight_decay_end if a.weight_decay_end is not None else self.weight_decay
        scale = global_batch_size / a.reference_batch_size
        self.base_lr = lr * (math.sqrt(scale) if a.lr_scale_method == "sqrt" else scale)
        self.warmup_steps = min(a.warmup_steps, int(total_steps * a.warmup_max_steps_fraction))
        dev, names = self.device, self.student.names
        self._untrained = untrained
        last = {"head." + WN_G, "head." + WN_V}
        self.groups = ["params_last_layer" if n in last else ("params" if decayed(n) else "params_no_weight_decay") for n in names]
        lr_live = [0.0 if n in self._untrained else self.base_lr for n in names]
        wd_live = [0 if (n in self._untrained or g_ == "params_no_weight_decay") else 1 for n, g_ in zip(names, self.groups)]
        mk = lambda v, dt: torch.tensor(v, dtype=dt, device=dev)
        self.seg_lr = mk(lr_live, torch.float32)
        self.seg_wd_on = mk(wd_live, torch.uint8)
        # the frozen last layer (dino.py:470-473): lr 0 and weight decay 0 for the group, everything else unchanged
        self.seg_lr_frozen = mk([0.0 if g_ == "params_last_layer" else v for v, g_ in zip(lr_live, self.groups)], torch.float32)
        self.seg_wd_on_frozen = mk([0 if g_ == "params_last_layer" else v for v, g_ in zip(wd_live, self.groups)], torch.uint8)
        self.seg_frozen = mk([1 if g_ == "params_last_layer" else 0 for g_ in self.groups], torch.uint8)
        self.param_groups = []   # (DINOv2's per-tensor AdamW groups do not apply)
        self.momentum_buffer = torch.zeros_like(self.student.data) if self.optimizer == "sgd" else None
        self.exp_avg = torch.zeros_like(self.student.data) if self.optimizer != "sgd" else None     # type: ignore[assignment]
        self.exp_avg_sq = torch.zeros_like(self.student.data) if self.optimizer != "sgd" else None  # type: ignore[assignment]
        self.opt_step = 0

    # ------------------------------------------------------------------ the step
    def training_step_impl(self, batch: Mapping[str, Any], batch_idx: int, masks: Any = None) -> TrainingStepResult:
        a, cfg, ws, dev = self.method_args, self.cfg, self.ws, self.device
        k, total = self.trainer.global_step, self.trainer.estimated_stepping_batches
        # EMA teacher first, momentum at this step (dino.py:273-284)
        momentum = cosine_schedule(k, total, a.momentum_start, a.momentum_end)
        ops.ema_flat(self.teacher.data, self.student.data, self.teacher.bf16, momentum)
        self.t_head.refresh_weightnorm()
        self.t_vit.refresh_padded_weights()
        teacher_temp = teacher_temp_schedule(a.teacher_temp, a.warmup_teacher_temp, a.warmup_teacher_temp_steps, k)

        views: List[Tensor] = [v.to(dev, torch.float32, non_blocking=True) for v in batch["views"]]
        n_views, n_local = len(views), len(views) - 2
        gv = torch.cat(views[:2])
        B = gv.shape[0] // 2
        p, n_reg = cfg.patch_size, cfg.num_register_tokens
        n_p = (-(-gv.shape[2] // p)) * (-(-gv.shape[3] // p))
        Ng = n_p + 1 + n_reg
        lv = torch.cat(views[2:]) if n_local > 0 else None
        n_p_l = (-(-lv.shape[2] // p)) * (-(-lv.shape[3] // p)) if lv is not None else 0
        Nl = n_p_l + 1 + n_reg
        D, K = cfg.embed_dim, a.output_dim
        ix = self._indices(B, n_p + n_reg, n_local, n_p_l + n_reg)
        if self._grad_sync is not None:
            self._grad_sync.reset()
        self.student.grad.zero_()
        self._loss_slots.zero_()
        rows_g = (ix["s_cls"], 2 * B) if self.sparse_last_mlp else None    # the loss reads the class-token rows only
        rows_l = (ix["l_cls"], n_local * B) if (self.sparse_last_mlp and n_local > 0) else None

        # ---------------- teacher (no grad) on its own stream: dino.py:326-331
        main = torch.cuda.current_stream()
        tstream = self.teacher_stream if (self.teacher_stream is not None and self.overlap_streams) else main
        tstream.wait_event(main.record_event())
        torch.cuda.set_stream(tstream)
        tctx = self.t_vit.forward(ws, "t", gv, None, save=False, last_mlp_rows=rows_g)
        t_in = ws.get("t.head_in", (2 * B, D), torch.bfloat16)
        ops.gather_rows(tctx["xn"].view(-1, D), D, ix["t_cls"], 2 * B, D, out_bf16=t_in)   # view halves swapped: row r faces student row r
        t_logits = self.t_head.forward(ws, "th", t_in, 2 * B, 2 * B, save=False)["logits"]
        t_probs = ws.get("t.probs", (2 * B, K), torch.float32)
        ops.softmax_center(t_logits, self.center.view(-1), t_probs, 2 * B, K, 1.0 / teacher_temp)
        # Center.update (lightly): center <- m center + (1 - m) mean over views, batch and ranks of the raw teacher outputs
        cs = ws.get("t.colsum", (K,), torch.float32)
        ops.colsum_f32(t_logits, cs, 2 * B, K)
        if self.world > 1:
            dist.all_reduce(cs)
        ops.center_ema(self.center.view(-1), cs, 1.0 / (2 * B * self.world), a.center_momentum, K)
        teacher_done = tstream.record_event()
        torch.cuda.set_stream(main)

        # ---------------- student: dino.py:333-337
        plan_g = batch.get("drop_plan_global", None) if isinstance(batch, dict) else None
        plan_l = batch.get("drop_plan_local", None) if isinstance(batch, dict) else None
        if plan_g is None:
            plan_g = make_drop_plan(cfg, 2 * B, self._drop_gen)
        if plan_l is None and lv is not None:
            plan_l = make_drop_plan(cfg, lv.shape[0], self._drop_gen)
        lstream = self.side_stream if (self.side_stream is not None and self.overlap_streams and lv is not None) else None
        sl = None
        if lstream is not None:
            lstream.wait_event(main.record_event())
            with torch.cuda.stream(lstream):
                sl = self.s_vit.forward(ws, "sl", lv, None, save=True, drop_plan=plan_l, checkpoint=self.activation_checkpointing, last_mlp_rows=rows_l)
                local_done = lstream.record_event()
        sg = self.s_vit.forward(ws, "sg", gv, None, save=True, drop_plan=plan_g, checkpoint=self.activation_checkpointing, last_mlp_rows=rows_g)
        if lstream is not None:
            main.wait_event(local_done)
        elif lv is not None:
            sl = self.s_vit.forward(ws, "sl", lv, None, save=True, drop_plan=plan_l, checkpoint=self.activation_checkpointing, last_mlp_rows=rows_l)
        Rl = n_local * B
        Rs = 2 * B + Rl
        s_in = ws.get("s.head_in", (Rs, D), torch.bfloat16, pad_rows=64)
        ops.gather_rows(sg["xn"].view(-1, D), D, ix["s_cls"], 2 * B, D, out_bf16=s_in[:2 * B])
        if sl is not None:
            ops.gather_rows(sl["xn"].view(-1, D), D, ix["l_cls"], Rl, D, out_bf16=s_in[2 * B:Rs])
        sh = self.s_head.forward(ws, "sh", s_in, Rs, Rs, save=True)

        # ---------------- DINOLoss: student global row r against teacher row r (the other view), local rows against both views
        n_terms = 2 * n_views - 2
        r2 = torch.arange(2 * B, dtype=torch.int32)
        bb = torch.arange(B, dtype=torch.int32)
        ta = torch.cat([r2, bb.repeat(n_local)])
        tb = torch.cat([torch.full((2 * B,), -1, dtype=torch.int32), (B + bb).repeat(n_local)])
        coef = torch.full((Rs,), 1.0 / (n_terms * B))
        slot = torch.cat([torch.zeros(2 * B, dtype=torch.int32), torch.ones(Rl, dtype=torch.int32)])
        ta, tb, coef, slot = (t.to(dev, non_blocking=True) for t in (ta, tb, coef, slot))
        main.wait_event(teacher_done)
        dlogits = ws.get("s.dlogits", (Rs, K), torch.bfloat16, pad_rows=64)
        ops.ce_fwd_bwd(sh["logits"], t_probs, ta, tb, coef, 1.0, 1.0 / a.student_temp, self._loss_slots, dlogits, Rs, K, slot=slot)

        # ---------------- backward
        self._reduce_begin()
        dx_head = self.s_head.backward(ws, sh, dlogits)
        self.s_head.finish_weightnorm_grad()
        if a.norm_last_layer:
            self._gwn.zero_()      # weight_g.requires_grad = False: no gradient, not part of the clipping norm
        dxn_g = ws.get("sg.dxn", (2 * B * Ng, D), torch.float32)
        dxn_g.zero_()
        ops.scatter_add_rows(dx_head[:2 * B], ix["s_cls"], dxn_g, D, 2 * B, D)
        dxn_l = None
        if sl is not None:
            dxn_l = ws.get("sl.dxn", (Rl * Nl, D), torch.float32)
            dxn_l.zero_()
            ops.scatter_add_rows(dx_head[2 * B:Rs], ix["l_cls"], dxn_l, D, Rl, D)
        self._backward_backbone(sg, dxn_g, sl, dxn_l)

        ls = self._loss_slots
        self._last = dict(t_logits=t_logits, t_probs=t_probs, s_global_logits=sh["logits"][:2 * B], s_local_logits=sh["logits"][2 * B:Rs], B=B, Rl=Rl)
        return TrainingStepResult(loss=ls[0] + ls[1], log_dict={"schedule/momentum": momentum, "schedule/teacher_temp": teacher_temp})

    # ------------------------------------------------------------------ optimizer hooks
    def _hparams_now(self) -> Dict[str, Any]:
        a, k, total = self.method_args, self.trainer.global_step, self.trainer.estimated_stepping_batches
        return dict(frozen=k < a.student_freeze_last_layer_steps, weight_decay=cosine_schedule(k, total, self.wd_start, self.wd_end),
                    lr_factor=warmup_cosine_lr_factor(k, self.warmup_steps, total, 0.001))   # CosineWarmupScheduler's default end value

    def optimizer_step(self) -> Dict[str, float]:
        """on_before_optimizer_step + clipping at 3.0 + SGD / AdamW + CosineWarmupScheduler (dino.py:330-340,354-477)."""
        a, h = self.method_args, self._hparams_now()
        self.allreduce_gradients()
        self._sumsq.zero_()
        ops.sumsq(self.student.grad, self._sumsq)
        self.opt_step += 1
        S = self.student
        if self.optimizer == "sgd":
            seg_lr, seg_wd = (self.seg_lr_frozen, self.seg_wd_on_frozen) if h["frozen"] else (self.seg_lr, self.seg_wd_on)
            ops.sgd_flat(S.data, S.grad, self.momentum_buffer, S.bf16, S.seg_of_chunk, seg_lr, seg_wd, h["lr_factor"], h["weight_decay"], a.momentum, 0.0,
                         False, self.opt_step == 1, self._sumsq, a.gradient_clip_val)
        else:
            ops.adamw_flat(S.data, S.grad, self.exp_avg, self.exp_avg_sq, S.bf16, S.seg_of_chunk, self.seg_lr, self.seg_wd_on, self.seg_frozen,
                           1 if h["frozen"] else 0, h["lr_factor"], h["weight_decay"], a.betas[0], a.betas[1], a.eps, self.opt_step, self._sumsq,
                           a.gradient_clip_val)
        self.s_head.refresh_weightnorm()
        self.s_vit.refresh_padded_weights()
        self.last_grad_norm = self._sumsq
        self.trainer.global_step += 1
        return {"weight_decay": h["weight_decay"], "lr_factor": h["lr_factor"]}

    def on_train_batch_end(self) -> float:   # the EMA runs at the start of training_step_impl in this method
        return 0.0

    # ------------------------------------------------------------------ reference-compatible views
    def _ref_key(self, role: str, flat: str) -> str:
        if flat.startswith("backbone."):
            return f"{role}_embedding_model.wrapped_model._model." + checkpoint.vit_key_from_flat(flat[9:], self.cfg.depth, self.cfg.block_chunks)
        return f"{role}_projection_head." + head_key_to_ref(flat[len("head."):])

    def state_dict(self) -> Dict[str, Tensor]:
        """`method.state_dict()` with the reference's keys (teacher first, as the modules are registered: dino.py:238-262)."""
        out: Dict[str, Tensor] = {}
        for role, fp in (("teacher", self.teacher), ("student", self.student)):
            for part in ("backbone.", "head."):
                for n in fp.names:
                    if n.startswith(part):
                        out[self._ref_key(role, n)] = fp.p[n].detach().clone()
        out["criterion.center.center"] = self.center.detach().clone().view(1, 1, -1)
        return out

    def load_state_dict(self, sd: Mapping[str, Tensor], strict: bool = True) -> None:
        want = {self._ref_key(role, n): (fp, n) for role, fp in (("teacher", self.teacher), ("student", self.student)) for n in fp.names}
        centers = ("criterion.center.center", "criterion.center")    # (older lightly releases register the buffer on the loss itself)
        missing = [k_ for k_ in want if k_ not in sd]
        unexpected = [k_ for k_ in sd if k_ not in want and k_ not in centers]
        if strict and (missing or unexpected or not any(c in sd for c in centers)):
            raise KeyError(f"load_state_dict: missing {missing[:5]}, unexpected {unexpected[:5]}")
        for k_, (fp, n) in want.items():     # validate everything before touching the storage
            if k_ in sd and tuple(sd[k_].shape) != tuple(fp.shapes[n]):
                raise ValueError(f"load_state_dict: {k_} has shape {tuple(sd[k_].shape)}, expected {tuple(fp.shapes[n])}")
        for k_, (fp, n) in want.items():
            if k_ in sd:
                fp.p[n].copy_(sd[k_].to(self.device, torch.float32))
        for fp in (self.student, self.teacher):
            fp.bf16.copy_(fp.data)
        for c in centers:
            if c in sd:
                self.center.copy_(sd[c].to(self.device, torch.float32).view_as(self.center))
                break
        self._refresh_derived()

    def _opt_order(self) -> List[Tuple[str, List[str]]]:
        """The reference optimizer's groups and, in each, the flat names in its parameter order (module traversal = named_parameters
        order; head after backbone; `params_last_layer` = last_layer.parameters() = weight_g, weight_v)."""
        order: Dict[str, List[str]] = {"params": [], "params_last_layer": [], "params_no_weight_decay": []}
        for n, g_ in zip(self.student.names, self.groups):
            order[g_].append(n)
        return list(order.items())

    def optimizer_state_dict(self) -> Dict[str, Any]:
        """`torch.optim.SGD.state_dict()` / `AdamW.state_dict()` of the reference's optimizer from the flat state."""
        a, h, S = self.method_args, self._hparams_now(), self.student
        lr_now = self.base_lr * h["lr_factor"]
        groups, state, idx = [], {}, 0
        for gname, names in self._opt_order():
            frozen_group = gname == "params_last_layer" and h["frozen"]
            wd = 0.0 if (gname == "params_no_weight_decay" or frozen_group) else (h["weight_decay"] if self.trainer.global_step > 0 else self.weight_decay)
            if self.optimizer == "sgd":
                hyper = dict(momentum=a.momentum, dampening=0, nesterov=False, maximize=False, foreach=None, differentiable=False, fused=None)
            else:
                hyper = dict(betas=tuple(a.betas), eps=a.eps, amsgrad=False, maximize=False, foreach=None, capturable=False, differentiable=False, fused=None,
                             decoupled_weight_decay=True)
            groups.append(dict(name=gname, lr=lr_now, weight_decay=wd, initial_lr=self.base_lr, params=list(range(idx, idx + len(names))), **hyper))
            for n in names:
                if self.opt_step > 0 and n not in self._untrained:
                    o, cnt, shape = S.offsets[n], S.p[n].numel(), S.shapes[n]
                    if self.optimizer == "sgd":
                        state[idx] = {"momentum_buffer": self.momentum_buffer[o:o + cnt].view(shape).detach().clone()}
                    else:
                        state[idx] = {"step": torch.tensor(float(self.opt_step)), "exp_avg": self.exp_avg[o:o + cnt].view(shape).detach().clone(),
                                      "exp_avg_sq": self.exp_avg_sq[o:o + cnt].view(shape).detach().clone()}
                idx += 1
        return {"state": state, "param_groups": groups}

    def load_optimizer_state_dict(self, osd: Mapping[str, Any]) -> None:
        S = self.student
        flat_names = [n for _, names in self._opt_order() for n in names]
        if sum(len(g_["params"]) for g_ in osd["param_groups"]) != len(flat_names):
            raise ValueError("optimizer state does not match this model's parameter count")
        bufs = [b for b in (self.momentum_buffer, self.exp_avg, self.exp_avg_sq) if b is not None]
        for b in bufs:
            b.zero_()          # parameters without an entry (never stepped) start from zero moments
        step = 0
        for i, n in enumerate(flat_names):
            st = osd["state"].get(i)
            if st is None:
                continue
            o, cnt = S.offsets[n], S.p[n].numel()
            if self.optimizer == "sgd":
                if st.get("momentum_buffer") is not None:
                    self.momentum_buffer[o:o + cnt].copy_(st["momentum_buffer"].reshape(-1).to(self.device, torch.float32))
                    step = max(step, 1)
            else:
                self.exp_avg[o:o + cnt].copy_(st["exp_avg"].reshape(-1).to(self.device, torch.float32))
                self.exp_avg_sq[o:o + cnt].copy_(st["exp_avg_sq"].reshape(-1).to(self.device, torch.float32))
                step = max(step, int(st["step"]))
        self.opt_step = step

    def _lr_end_value(self) -> float:
        return 0.001       # CosineWarmupScheduler's default (dino.py:404-413 passes none)

    def _opt_counts_steps(self) -> bool:
        return self.optimizer != "sgd"     # torch's SGD keeps no step count

    def load_checkpoint_dict(self, ckpt: Mapping[str, Any], strict: bool = True) -> None:
        super().load_checkpoint_dict(ckpt, strict)
        if self.optimizer == "sgd" and self.trainer.global_step > 0:
            self.opt_step = self.trainer.global_step      # the momentum buffers exist from the first step on

    def train_step(self, views: List[Tensor], masks: Any = None) -> TrainingStepResult:
        res = self.training_step_impl({"views": views}, 0)
        self.optimizer_step()
        return res


class DINOResNet(DINO):
    """DINO on a torchvision ResNet (the reference's `ResNetModelWrapper`: `_features` = conv1 .. layer4, pooled by `_pool`; BASELINE's
    classic `model="torchvision/resnet50"` pairing).  Same step as `DINO`; what the backbone changes:

      * student AND teacher run BatchNorm in training mode (the reference never puts the teacher in eval()): batch statistics per forward
        call -- the student's global and local views are two calls -- and every call moves that network's running estimates; the EMA walks
        parameters only, so the teacher's BatchNorm buffers are its own;
      * `EmbeddingModel(x)` is the average-pooled layer4 map (`lt_token_mean_bf16` / `lt_pool_bwd_add`);
      * weight decay: everything that is not a BatchNorm parameter or a bias, i.e. the convolution and Linear weights.

    Kernels: `resnet.ResNetEngine` (NHWC, every convolution an MFMA GEMM, `csrc/conv.hip`) + the projection-head / loss / optimizer kernels
    of the ViT path.  Convolutional weight gradients keep the dispatcher's split-K choice: reproducible to the last bits."""

    def __init__(self, cfg: Any, method_args: Optional[DINOArgs] = None, global_batch_size: int = 256, total_steps: int = 125_000,
                 device: "str | torch.device" = "cuda", backbone_state: Optional[Dict[str, Tensor]] = None,
                 student_head_state: Optional[Dict[str, Tensor]] = None, teacher_head_state: Optional[Dict[str, Tensor]] = None,
                 teacher_backbone_state: Optional[Dict[str, Tensor]] = None, seed: int = 0) -> None:
        from .dinov2 import HeadEngine, MockTrainerState, head_param_shapes, init_head_state
        from .params import FlatParams
        from .resnet import ResNetEngine, flat_named, init_resnet_state
        from .vit import Workspace

        a = method_args or DINOArgs()
        if a.batch_norm:
            raise NotImplementedError("DINO(batch_norm=True): lightly's shared-BatchNorm1d head is not built")
        if a.optimizer not in ("auto", "sgd", "adamw"):
            raise ValueError(f"Invalid optimizer type: '{a.optimizer}'")
        self.method_args, self.cfg = a, cfg     # type: ignore[assignment]
        self.device = torch.device(device)
        self.global_batch_size = global_batch_size
        self.trainer = MockTrainerState(total_steps)
        g = torch.Generator().manual_seed(seed)
        D = cfg.feature_dim
        bsd = backbone_state if backbone_state is not None else init_resnet_state(cfg, g)
        tbs = teacher_backbone_state if teacher_backbone_state is not None else bsd
        conv = (lambda s_: None if s_ is None else head_state_from_ref(s_) if any(k.startswith("layers.") for k in s_) else s_)
        shs = conv(student_head_state) or init_head_state(D, a.hidden_dim, a.bottleneck_dim, a.output_dim, g)
        ths = conv(teacher_head_state) or init_head_state(D, a.hidden_dim, a.bottleneck_dim, a.output_dim, g)
        order_h = [n for n, _ in head_param_shapes(D, a.hidden_dim, a.bottleneck_dim, a.output_dim)]
        self.student = FlatParams(flat_named(cfg, bsd, "backbone.") + [("head." + n, shs[n]) for n in order_h], self.device, True)
        self.teacher = FlatParams(flat_named(cfg, tbs, "backbone.") + [("head." + n, ths[n]) for n in order_h], self.device, False)
        if self.world > 1:
            for fp in (self.student, self.teacher):
                dist.broadcast(fp.data, src=0)
                fp.bf16.copy_(fp.data)
        self.s_net = ResNetEngine(cfg, self.student, "backbone.", buffers=bsd)
        self.t_net = ResNetEngine(cfg, self.teacher, "backbone.", buffers=tbs)
        self.s_vit = self.s_net            # (the inherited optimizer step refreshes the derived weight copies through this name)
        shim = DINOv2Args(hidden_dim=a.hidden_dim, dino_bottleneck_dim=a.bottleneck_dim, output_dim=a.output_dim, batch_norm=False)
        self.s_head = HeadEngine(self.student, "head.", D, shim)
        self.t_head = HeadEngine(self.teacher, "head.", D, shim)
        self.s_head.refresh_weightnorm()
        self.t_head.refresh_weightnorm()
        self.center = self.dino_center = torch.zeros(1, a.output_dim, device=self.device)
        self.ws = Workspace(self.device)
        dev = self.device
        shapes = self.student.shapes
        self._setup_optimizer(global_batch_size, total_steps, untrained=({"head." + WN_G} if a.norm_last_layer else set()),
                              decayed=lambda n: len(shapes[n]) > 1 and not n.endswith("bias"))    # convolution / Linear weights
        self._sumsq = torch.zeros(1, device=dev)
        self._loss_slots = torch.zeros(5, device=dev)
        self._gwn = self.student.g["head." + WN_G]
        self._pending = {}
        self._grad_sync = None
        self.comm_events = None
        self.last_grad_norm = None
        self.overlap_streams = True
        use_streams = self.device.type == "cuda"
        self.side_stream = torch.cuda.Stream(device=self.device) if use_streams else None
        self.teacher_stream = torch.cuda.Stream(device=self.device) if use_streams else None

    # ------------------------------------------------------------------ the step
    def training_step_impl(self, batch: Mapping[str, Any], batch_idx: int, masks: Any = None) -> TrainingStepResult:
        a, ws, dev = self.method_args, self.ws, self.device
        k, total = self.trainer.global_step, self.trainer.estimated_stepping_batches
        momentum = cosine_schedule(k, total, a.momentum_start, a.momentum_end)
        ops.ema_flat(self.teacher.data, self.student.data, self.teacher.bf16, momentum)
        self.t_head.refresh_weightnorm()
        self.t_net.refresh_padded_weights()
        teacher_temp = teacher_temp_schedule(a.teacher_temp, a.warmup_teacher_temp, a.warmup_teacher_temp_steps, k)
        views: List[Tensor] = [v.to(dev, torch.float32, non_blocking=True) for v in batch["views"]]
        n_views, n_local = len(views), len(views) - 2
        gv = torch.cat(views[:2])
        B = gv.shape[0] // 2
        lv = torch.cat(views[2:]) if n_local > 0 else None
        D, K = self.cfg.feature_dim, a.output_dim
        Rl = n_local * B
        Rs = 2 * B + Rl
        if self._grad_sync is not None:
            self._grad_sync.reset()
        self.student.grad.zero_()
        self._loss_slots.zero_()

        # ---------------- teacher (no grad; BatchNorm in training mode) on its own stream
        main = torch.cuda.current_stream()
        tstream = self.teacher_stream if (self.teacher_stream is not None and self.overlap_streams) else main
        tstream.wait_event(main.record_event())
        torch.cuda.set_stream(tstream)
        tc = self.t_net.forward(ws, "t", gv, save=False, train=True)
        t_in = ws.get("t.head_in", (2 * B, D), torch.bfloat16)
        ops.token_mean(tc["feat"], t_in, 2 * B, tc["h"] * tc["w"], D)
        t_logits = self.t_head.forward(ws, "th", t_in, 2 * B, 2 * B, save=False)["logits"]
        t_probs = ws.get("t.probs", (2 * B, K), torch.float32)
        ops.softmax_center(t_logits, self.center.view(-1), t_probs, 2 * B, K, 1.0 / teacher_temp)
        cs = ws.get("t.colsum", (K,), torch.float32)
        ops.colsum_f32(t_logits, cs, 2 * B, K)
        if self.world > 1:
            dist.all_reduce(cs)
        ops.center_ema(self.center.view(-1), cs, 1.0 / (2 * B * self.world), a.center_momentum, K)
        teacher_done = tstream.record_event()
        torch.cuda.set_stream(main)

        # ---------------- student: two forward calls (global, local), pooled, one projection-head batch
        s_in = ws.get("s.head_in", (Rs, D), torch.bfloat16, pad_rows=64)
        sg = self.s_net.forward(ws, "sg", gv, save=True, train=True)
        ops.token_mean(sg["feat"], s_in[:2 * B], 2 * B, sg["h"] * sg["w"], D)
        sl = None
        if lv is not None:
            sl = self.s_net.forward(ws, "sl", lv, save=True, train=True)
            ops.token_mean(sl["feat"], s_in[2 * B:Rs], Rl, sl["h"] * sl["w"], D)
        sh = self.s_head.forward(ws, "sh", s_in, Rs, Rs, save=True)

        # ---------------- DINOLoss: student global row r against the teacher's other view (r + B) mod 2B, local rows against both
        n_terms = 2 * n_views - 2
        r2 = torch.arange(2 * B, dtype=torch.int32)
        bb = torch.arange(B, dtype=torch.int32)
        ta = torch.cat([(r2 + B) % (2 * B), bb.repeat(n_local)])
        tb = torch.cat([torch.full((2 * B,), -1, dtype=torch.int32), (B + bb).repeat(n_local)])
        coef = torch.full((Rs,), 1.0 / (n_terms * B))
        slot = torch.cat([torch.zeros(2 * B, dtype=torch.int32), torch.ones(Rl, dtype=torch.int32)])
        ta, tb, coef, slot = (t.to(dev, non_blocking=True) for t in (ta, tb, coef, slot))
        main.wait_event(teacher_done)
        dlogits = ws.get("s.dlogits", (Rs, K), torch.bfloat16, pad_rows=64)
        ops.ce_fwd_bwd(sh["logits"], t_probs, ta, tb, coef, 1.0, 1.0 / a.student_temp, self._loss_slots, dlogits, Rs, K, slot=slot)

        # ---------------- backward: head, then the two convolutional passes (weight gradients on the side stream)
        dx_head = self.s_head.backward(ws, sh, dlogits)
        self.s_head.finish_weightnorm_grad()
        if a.norm_last_layer:
            self._gwn.zero_()
        side = self.side_stream if self.overlap_streams else None
        for tag, ctx, lo, hi in (("sg", sg, 0, 2 * B), ("sl", sl, 2 * B, Rs)):
            if ctx is None:
                continue
            n = ctx["h"] * ctx["w"]
            dfeat = ws.get(tag + ".dfeat", (ctx["feat"].shape[0], D), torch.bfloat16, zero=True)
            ops.pool_bwd_add(None, dx_head[lo:hi], dfeat, hi - lo, n, D)
            self.s_net.backward(ws, ctx, dfeat, side=side)
        if side is not None:
            main.wait_stream(side)
        ls = self._loss_slots
        self._last = dict(t_logits=t_logits, t_probs=t_probs, s_global_logits=sh["logits"][:2 * B], s_local_logits=sh["logits"][2 * B:Rs], B=B, Rl=Rl)
        return TrainingStepResult(loss=ls[0] + ls[1], log_dict={"schedule/momentum": momentum, "schedule/teacher_temp": teacher_temp})

    def _refresh_derived(self) -> None:
        self.s_head.refresh_weightnorm()
        self.t_head.refresh_weightnorm()
        self.s_net.refresh_padded_weights()
        self.t_net.refresh_padded_weights()

    # ------------------------------------------------------------------ reference-compatible views
    def state_dict(self) -> Dict[str, Tensor]:
        out: Dict[str, Tensor] = {}
        for role, fp, net in (("teacher", self.teacher, self.t_net), ("student", self.student, self.s_net)):
            for k_, v in net.state_dict().items():
                out[f"{role}_embedding_model.wrapped_model._features.{k_}"] = v
            for n in fp.names:
                if n.startswith("head."):
                    out[f"{role}_projection_head." + head_key_to_ref(n[5:])] = fp.p[n].detach().clone()
        out["criterion.center.center"] = self.center.detach().clone().view(1, 1, -1)
        return out

    def load_state_dict(self, sd: Mapping[str, Tensor], strict: bool = True) -> None:
        from .resnet import resnet_param_shapes, to_flat_layout

        plan: List[Tuple[Tensor, Tensor]] = []
        bufs: List[Tuple[Any, Dict[str, Tensor]]] = []
        seen = set()
        for role, fp, net in (("teacher", self.teacher, self.t_net), ("student", self.student, self.s_net)):
            pre = f"{role}_embedding_model.wrapped_model._features."
            bb = {k_[len(pre):]: v for k_, v in sd.items() if k_.startswith(pre)}
            for n, shape in resnet_param_shapes(self.cfg):
                if n in bb:
                    if tuple(bb[n].shape) != tuple(shape):
                        raise ValueError(f"load_state_dict: {pre}{n} has shape {tuple(bb[n].shape)}, expected {tuple(shape)}")
                    plan.append((fp.p["backbone." + n], to_flat_layout(n, bb[n].float())))
                    seen.add(pre + n)
                elif strict:
                    raise KeyError(f"load_state_dict: missing {pre}{n}")
            for k_ in net.buffers:
                if k_ in bb:
                    seen.add(pre + k_)
                elif strict:
                    raise KeyError(f"load_state_dict: missing BatchNorm buffer {pre}{k_}")
            bufs.append((net, bb))
            hp = f"{role}_projection_head."
            for n in fp.names:
                if n.startswith("head."):
                    key = hp + head_key_to_ref(n[5:])
                    if key in sd:
                        if tuple(sd[key].shape) != tuple(fp.shapes[n]):
                            raise ValueError(f"load_state_dict: {key} has shape {tuple(sd[key].shape)}, expected {tuple(fp.shapes[n])}")
                        plan.append((fp.p[n], sd[key]))
                        seen.add(key)
                    elif strict:
                        raise KeyError(f"load_state_dict: missing {key}")
        centers = ("criterion.center.center", "criterion.center")
        extra = [k_ for k_ in sd if k_ not in seen and k_ not in centers and not (".fc." in k_)]
        if strict and (extra or not any(c in sd for c in centers)):
            raise KeyError(f"load_state_dict: unexpected {extra[:5]} / no center")
        for dst, src in plan:
            dst.copy_(src.to(self.device, torch.float32))
        for net, bb in bufs:
            net.load_buffers(bb)
        for fp in (self.student, self.teacher):
            fp.bf16.copy_(fp.data)
        for c in centers:
            if c in sd:
                self.center.copy_(sd[c].to(self.device, torch.float32).view_as(self.center))
                break
        self._refresh_derived()

    def _ref_key(self, role: str, flat: str) -> str:
        if flat.startswith("backbone."):
            return f"{role}_embedding_model.wrapped_model._features." + flat[9:]
        return f"{role}_projection_head." + head_key_to_ref(flat[len("head."):])

    def export_backbone_state_dict(self) -> Dict[str, Tensor]:
        return self.t_net.state_dict()
