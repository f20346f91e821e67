"""Distillation (v1) and DistillationV2 on the MI355X kernels -- the two siblings of DistillationV3.

  * `Distillation`   mirrors LT/_methods/distillation/distillation.py:155-362 + distillation_loss.py:16-75: one mixed-up view, frozen
    ViT teacher -> pooled (cls) feature, student pooled feature -> `student_projection_head = Linear(embed_dim, teacher_dim)` -> both
    L2-normalised -> similarities against a queue of past teacher features -> KL(softmax(t / T) || softmax(s / T)), "batchmean".
  * `DistillationV2` mirrors LT/_methods/distillationv2/distillationv2.py:156-377 + distillationv2_loss.py:14-44: teacher features =
    channel-concatenation of the last `n_teacher_blocks` blocks' normed patch tokens (`get_intermediate_layers(x, n, reshape=True)`),
    student feature map -> `student_projection_head.mlp` (a Linear for `n_projection_layers = 1`) -> bilinear resize onto the teacher
    grid -> MSE over all elements.
Both train with gradient-clip 1.0 and the generic `Method.configure_optimizers` schedule (method.py:89-121).  Optimizers: the default
`optimizer="auto"` is the reference's "auto" = LARS (`Distillation(V2)LARSArgs`: lr 1.8, momentum 0.9, weight decay 1e-6;
distillation.py:140-147,290-298, distillationv2.py:106-113,306-314) on `lars.FlatLARS`; `optimizer="adamw"` selects v1's
`DistillationAdamWArgs` (lr 5e-4, weight decay 0) / the generic `AdamWArgs` for v2 (lr 1e-3, weight decay 0.01), whose values are the
`lr` / `weight_decay` fields below.

Students: a ViT on `vit.ViTEngine` or the torchvision ResNet on `resnet.ResNetEngine`; teacher: a DINOv2 / DINOv3 ViT.  State-dict
names follow the reference (`student_embedding_model.wrapped_model.*`, `student_projection_head.*`, `teacher_queue`)."""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Any, Dict, List, Mapping, Optional, Tuple

import torch
import torch.distributed as dist
from torch import Tensor

from . import ops
from .distillationv3 import TrainingStepResult, _Trainer, weight_decays
from .lars import FlatLARS, LARSArgs
from .parallel import GradSync
from .params import FlatParams
from .resnet import ResNetConfig, ResNetEngine, flat_named, init_resnet_state
from .schedules import warmup_cosine_lr_factor
from .vit import ViTConfig, ViTEngine, Workspace, _split_k, init_vit_state, split_k_plan, vit_param_shapes


@dataclass
class DistillationArgs:
    """DistillationArgs (distillation.py:81-137) + DistillationAdamWArgs (:150-152)."""
    queue_size: int = 8192
    temperature: float = 0.07
    lr_scale_method: str = "sqrt"
    reference_batch_size: int = 1536
    lr: float = 0.0005
    betas: Tuple[float, float] = (0.9, 0.999)
    eps: float = 1e-8
    weight_decay: float = 0.0
    gradient_clip_val: float = 1.0
    optimizer: str = "auto"           # "auto" = "lars" as in the reference (DistillationLARSArgs, distillation.py:140-147); "adamw": the fields above
    lars: LARSArgs = field(default_factory=LARSArgs)


@dataclass
class DistillationV2Args:
    """DistillationV2Args (distillationv2.py:81-105) + the generic AdamWArgs (LT/_optim/adamw_args.py:21-27)."""
    n_teacher_blocks: int = 2
    n_projection_layers: int = 1
    projection_hidden_dim: int = 2048
    lr_scale_method: str = "sqrt"
    reference_batch_size: int = 1536
    lr: float = 0.001
    betas: Tuple[float, float] = (0.9, 0.999)
    eps: float = 1e-8
    weight_decay: float = 0.01
    gradient_clip_val: float = 1.0
    optimizer: str = "auto"           # "auto" = "lars" as in the reference (DistillationV2LARSArgs, distillationv2.py:106-113); "adamw": the fields above
    lars: LARSArgs = field(default_factory=LARSArgs)


class _Student:
    """The student behind `EmbeddingModel(x, pool=...)`: pooled feature [B, D] (cls token / average pool) and feature-map tokens
    [B * n, D] (bf16), with the backward that takes the two gradients."""

    def __init__(self, cfg: "ViTConfig | ResNetConfig", params: FlatParams, state: Dict[str, Tensor]) -> None:
        self.cfg, self.P = cfg, params
        self.conv = isinstance(cfg, ResNetConfig)
        self.dim = cfg.feature_dim if self.conv else cfg.embed_dim
        self.net = ResNetEngine(cfg, params, "backbone.", buffers=state) if self.conv else ViTEngine(cfg, params, "backbone.")
        self._idx: Dict[Tuple[int, int, int], Tuple[Tensor, Tensor]] = {}

    def _rows(self, B: int, N: int, prefix: int) -> Tuple[Tensor, Tensor]:
        key = (B, N, prefix)
        if key not in self._idx:
            r = torch.arange(B, dtype=torch.int64)
            patch = (r[:, None] * N + torch.arange(prefix, N, dtype=torch.int64)[None, :]).reshape(-1)
            self._idx[key] = ((r * N).to(self.P.device), patch.to(self.P.device))
        return self._idx[key]

    def forward(self, ws: Workspace, x: Tensor, want_pooled: bool, want_tokens: bool) -> Dict[str, Any]:
        B, D = x.shape[0], self.dim
        if self.conv:
            sc = self.net.forward(ws, "s", x, save=True, train=True)
            n = sc["h"] * sc["w"]
            out: Dict[str, Any] = dict(ctx=sc, gh=sc["h"], gw=sc["w"], n=n, tokens=sc["feat"], pooled=None)
            if want_pooled:
                out["pooled"] = ops.token_mean(sc["feat"], ws.get("s.pool", (B, D), torch.bfloat16), B, n, D)
            return out
        sc = self.net.forward(ws, "s", x, None, save=True)
        N, pre = sc["N"], 1 + self.cfg.num_register_tokens
        n = N - pre
        cls_rows, patch_rows = self._rows(B, N, pre)
        xn = sc["xn"].view(-1, D)
        out = dict(ctx=sc, gh=sc["gh"], gw=sc["gw"], n=n, N=N, cls_rows=cls_rows, patch_rows=patch_rows, pooled=None, tokens=None)
        if want_pooled:
            out["pooled"] = ws.get("s.pool", (B, D), torch.bfloat16)
            ops.gather_rows(xn, D, cls_rows, B, D, out_bf16=out["pooled"])
        if want_tokens:
            out["tokens"] = ws.get("s.tok", (B * n, D), torch.bfloat16)
            ops.gather_rows(xn, D, patch_rows, B * n, D, out_bf16=out["tokens"])
        return out

    def backward(self, ws: Workspace, f: Dict[str, Any], B: int, d_pooled: Optional[Tensor], d_tokens: Optional[Tensor],
                 side: Optional["torch.cuda.Stream"]) -> None:
        D, n = self.dim, f["n"]
        if self.conv:
            dfeat = ws.get("s.dfeat", (f["ctx"]["feat"].shape[0], D), torch.bfloat16, zero=True)
            ops.pool_bwd_add(d_tokens, d_pooled, dfeat, B, n, D)
            self.net.backward(ws, f["ctx"], dfeat, side=side)
        else:
            N = f["N"]
            dxn = ws.get("s.dxn", (B * N, D), torch.float32)
            dxn.zero_()
            if d_pooled is not None:
                ops.scatter_add_rows(d_pooled, f["cls_rows"], dxn, D, B, D)
            if d_tokens is not None:
                ops.scatter_add_rows(d_tokens, f["patch_rows"], dxn, D, B * n, D)
            self.net.backward(ws, f["ctx"], dxn.view(B, N, D), side=side)
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)
        if not self.conv:
            self.net.finish_layerscale_grads()


class _DistillBase:
    HEAD: str = "student_projection_head."

    def __init__(self, student_cfg: "ViTConfig | ResNetConfig", teacher_cfg: ViTConfig, args: Any, head_shapes: List[Tuple[str, Tuple[int, ...]]],
                 global_batch_size: int, total_steps: int, max_epochs: int, device: "str | torch.device",
                 student_state: Optional[Dict[str, Tensor]], teacher_state: Optional[Dict[str, Tensor]], head_state: Optional[Dict[str, Tensor]],
                 seed: int) -> None:
        self.method_args = a = args
        self.scfg, self.tcfg = student_cfg, teacher_cfg
        self.device = dev = torch.device(device)
        ops.require_device(dev, type(self).__name__)
        g = torch.Generator().manual_seed(seed)
        conv = isinstance(student_cfg, ResNetConfig)
        sb = student_state if student_state is not None else (init_resnet_state(student_cfg, g) if conv else init_vit_state(student_cfg, g))
        tb = teacher_state if teacher_state is not None else init_vit_state(teacher_cfg, g)
        Ds = student_cfg.feature_dim if conv else student_cfg.embed_dim
        hs: Dict[str, Tensor] = {}
        for n, shape in head_shapes:
            if head_state is not None:
                hs[n] = head_state[n].detach().clone().float()
            elif len(shape) == 2:   # trunc_normal(std 0.02) weights (distillation.py:186, distillationv2.py:139-143)
                hs[n] = torch.nn.init.trunc_normal_(torch.empty(shape), std=0.02, generator=g)
            else:
                hs[n] = self._init_bias(shape, Ds, g)
        named = (flat_named(student_cfg, sb, "backbone.") if conv else [("backbone." + n, sb[n]) for n, _ in vit_param_shapes(student_cfg)])
        named += [("head." + n, hs[n]) for n, _ in head_shapes]
        self.student = FlatParams(named, dev, True)
        self.teacher = FlatParams([(n, tb[n]) for n, _ in vit_param_shapes(teacher_cfg)], dev, False)
        self.s = _Student(student_cfg, self.student, sb)
        self.t_vit = ViTEngine(teacher_cfg, self.teacher, "")
        self._fc = {k: sb[k].detach().clone() for k in ("fc.weight", "fc.bias") if conv and k in sb}
        self.ws = Workspace(dev)
        self.global_batch_size = global_batch_size
        self.trainer = _Trainer(total_steps, max_epochs)
        scale = global_batch_size / a.reference_batch_size
        if a.lr_scale_method == "sqrt":
            scale = math.sqrt(scale)
        if a.optimizer not in ("auto", "adamw", "lars"):
            raise ValueError(f"Invalid optimizer type: '{a.optimizer}'")
        self.optimizer = "lars" if a.optimizer == "auto" else a.optimizer   # both methods map "auto" to their LARS arguments
        self.base_lr = (a.lars.lr if self.optimizer == "lars" else a.lr) * scale
        warm_epochs = min(10, max(1, max_epochs) / 10)
        self.warmup_steps = min(int(total_steps), int(total_steps / max(1, max_epochs) * warm_epochs))
        self.lars = FlatLARS(self.student, a.lars) if self.optimizer == "lars" else None
        self.exp_avg = torch.zeros_like(self.student.data) if self.lars is None else None
        self.exp_avg_sq = torch.zeros_like(self.student.data) if self.lars is None else None
        nn_ = len(self.student.names)
        self.seg_lr = torch.full((nn_,), self.base_lr, dtype=torch.float32, device=dev)
        self.seg_wd_on = torch.tensor([1 if weight_decays(n, self.student.shapes[n]) else 0 for n in self.student.names], dtype=torch.uint8, device=dev)
        self.seg_frozen = torch.zeros(nn_, dtype=torch.uint8, device=dev)
        self._sumsq = torch.zeros(1, device=dev)
        self._loss = torch.zeros(1, device=dev)
        self.opt_step = 0
        self.last_grad_norm: Optional[Tensor] = None
        self._grad_sync: Optional[GradSync] = None
        self.teacher_stream = torch.cuda.Stream(device=dev)
        self.side_stream = torch.cuda.Stream(device=dev)

    @staticmethod
    def _init_bias(shape: Tuple[int, ...], fan_in: int, g: torch.Generator) -> Tensor:
        return torch.zeros(shape)

    @property
    def world(self) -> int:
        return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1

    def _mixup(self, batch: Mapping[str, Any], mix: Optional[Tuple[float, Tensor]]) -> Tuple[Tensor, float, Tensor]:
        """_mixup_data (distillation.py:268-279): lambda ~ U(0, 1), random permutation -- same host RNG draws in the same order."""
        views = batch["views"][0].to(self.device, torch.float32, non_blocking=True).contiguous()
        B = views.shape[0]
        if mix is None:
            lam = torch.empty(1).uniform_(0.0, 1.0).item()
            index = torch.randperm(B)
        else:
            lam, index = mix
        x = self.ws.get("mix.x", tuple(views.shape), torch.float32)
        ops.mixup(views, index.to(self.device, torch.int64), float(lam), x)
        return x, float(lam), index

    def _head_wgrad(self, name: str, dy: Tensor, xin: Tensor, n_out: int, k_in: int, rows: int) -> None:
        P = self.student
        slab = self.ws.get("wgrad.slabs", (32 * 1024 * 1024,), torch.float32)
        ops.colsum_bf16(dy, P.g["head." + name + "bias"], rows, n_out)
        tiles = ((n_out + 127) // 128) * ((k_in + 127) // 128)
        ops.gemm(dy, xin, P.g["head." + name + "weight"], M=n_out, N=k_in, K=rows, trans_a=True, trans_b=True, epilogue=ops.EPI_F32_ACCUM,
                 lda=n_out, ldb=k_in, workspace=slab, **split_k_plan(n_out, k_in, rows, True, _split_k(tiles, rows)))

    def optimizer_step(self) -> None:
        a = self.method_args
        k = self.trainer.global_step
        total = int(self.trainer.estimated_stepping_batches)
        lr_factor = warmup_cosine_lr_factor(k, self.warmup_steps, total, 0.001)   # CosineWarmupScheduler default end_value
        if self.world > 1:
            if self._grad_sync is None:
                self._grad_sync = GradSync(self.student.grad)
            self._grad_sync.finish()
        self._sumsq.zero_()
        ops.sumsq(self.student.grad, self._sumsq)
        self.opt_step += 1
        if self.lars is not None:
            self.lars.step(self.seg_lr, self.seg_wd_on, lr_factor, self._sumsq, a.gradient_clip_val)
        else:
            ops.adamw_flat(self.student.data, self.student.grad, self.exp_avg, self.exp_avg_sq, self.student.bf16, self.student.seg_of_chunk,
                           self.seg_lr, self.seg_wd_on, self.seg_frozen, 0, lr_factor, a.weight_decay, a.betas[0], a.betas[1], a.eps,
                           self.opt_step, self._sumsq, a.gradient_clip_val)
        self.s.net.refresh_padded_weights()
        self.last_grad_norm = self._sumsq
        self.trainer.global_step += 1

    def train_step(self, views: Tensor, mix: Optional[Tuple[float, Tensor]] = None) -> TrainingStepResult:
        res = self.training_step_impl({"views": [views]}, 0, mix=mix)
        self.optimizer_step()
        return res

    def _backbone_state(self) -> Dict[str, Tensor]:
        if self.s.conv:
            return {"student_embedding_model.wrapped_model._features." + k: v for k, v in self.s.net.state_dict().items()}
        from .checkpoint import vit_key_from_flat

        cfg = self.s.net.cfg     # chunked students (block_chunks > 0: vitl14 / vitg14) keep the reference's `blocks.<chunk>.<i>.` names
        return {"student_embedding_model.wrapped_model._model." + vit_key_from_flat(n[9:], cfg.depth, cfg.block_chunks): self.student.p[n].detach().clone()
                for n in self.student.names if n.startswith("backbone.")}

    def load_state_dict(self, sd: Mapping[str, Tensor], strict: bool = True) -> None:
        """Load what `state_dict()` (or the reference's Distillation / DistillationV2 `state_dict()`) wrote."""
        from . import checkpoint
        checkpoint.distill_load_state_dict(self, sd, {"student_projection_head.": "head."}, strict)

    def optimizer_state(self) -> Dict[str, Any]:
        from . import checkpoint
        return checkpoint.distill_optimizer_state(self)

    def load_optimizer_state(self, st: Mapping[str, Any]) -> None:
        from . import checkpoint
        checkpoint.distill_load_optimizer_state(self, st)

    def export_backbone_state_dict(self) -> Dict[str, Tensor]:
        if self.s.conv:
            return self.s.net.state_dict(extra=self._fc)
        return {n[9:]: self.student.p[n].detach().clone() for n in self.student.names if n.startswith("backbone.")}


class Distillation(_DistillBase):
    def __init__(self, student_cfg: "ViTConfig | ResNetConfig", teacher_cfg: ViTConfig, method_args: Optional[DistillationArgs] = None,
                 global_batch_size: int = 128, total_steps: int = 100_000, max_epochs: int = 100, device: "str | torch.device" = "cuda",
                 student_state: Optional[Dict[str, Tensor]] = None, teacher_state: Optional[Dict[str, Tensor]] = None,
                 head_state: Optional[Dict[str, Tensor]] = None, seed: int = 0) -> None:
        a = method_args or DistillationArgs()
        Ds = student_cfg.feature_dim if isinstance(student_cfg, ResNetConfig) else student_cfg.embed_dim
        Dt = teacher_cfg.embed_dim
        super().__init__(student_cfg, teacher_cfg, a, [("weight", (Dt, Ds)), ("bias", (Dt,))], global_batch_size, total_steps, max_epochs, device,
                         student_state, teacher_state, head_state, seed)
        self.teacher_queue = torch.zeros(a.queue_size, Dt, device=self.device)

    @staticmethod
    def _init_bias(shape: Tuple[int, ...], fan_in: int, g: torch.Generator) -> Tensor:
        bound = 1.0 / math.sqrt(fan_in)     # nn.Linear's default bias init (only the weight is re-initialised, distillation.py:186)
        return torch.empty(shape).uniform_(-bound, bound, generator=g)

    def training_step_impl(self, batch: Mapping[str, Any], batch_idx: int, mix: Optional[Tuple[float, Tensor]] = None) -> TrainingStepResult:
        a, ws, P = self.method_args, self.ws, self.student
        x, lam, index = self._mixup(batch, mix)
        B, Dt, Ds = x.shape[0], self.tcfg.embed_dim, self.s.dim
        main = torch.cuda.current_stream()
        if self._grad_sync is not None:
            self._grad_sync.reset()
        P.grad.zero_()
        self._loss.zero_()
        Q = self.teacher_queue.shape[0]
        Qp = (Q + 7) // 8 * 8
        ts = self.teacher_stream
        ts.wait_event(main.record_event())
        with torch.cuda.stream(ts):   # _forward_teacher (distillation.py:243-251) + _update_queue (:225-241)
            tc = self.t_vit.forward(ws, "t", x, None, save=False)
            rows = (torch.arange(B, dtype=torch.int64) * tc["N"]).to(self.device)
            tg_raw = ws.get("t.g", (B, Dt), torch.float32)
            ops.gather_rows(tc["xn"].view(-1, Dt), Dt, rows, B, Dt, out_f32=tg_raw)
            tg = ws.get("t.gn", (B, Dt), torch.bfloat16)
            tinv = ws.get("t.inv", (B,), torch.float32)
            ops.l2norm_fwd(tg_raw, tg, tinv, B, Dt, 1e-12)
            tgn = ws.get("t.gn32", (B, Dt), torch.float32)
            torch.mul(tg_raw, tinv[:, None], out=tgn)          # plumbing: fp32 rows for the queue
            if B >= Q:
                self.teacher_queue.copy_(tgn[:Q])
            else:
                self.teacher_queue[B:] = self.teacher_queue[:-B].clone()
                self.teacher_queue[:B] = tgn
            # similarity matrices [B, Q] are held with rows of Qp = Q rounded up to 8 columns (16-byte rows for the GEMMs); the pad rows of
            # the bf16 queue copy stay zero, the pad columns of the logits are never read (lt_kl_fwd_bwd takes the row stride)
            qb = ws.get("queue.bf16", (Qp, Dt), torch.bfloat16, zero=True)
            ops.cast_bf16(self.teacher_queue, qb[:Q])
            t_logits = ws.get("g.t_logits", (B, Qp), torch.float32)
            ops.gemm(tg, qb, t_logits, M=B, N=Qp, K=Dt, epilogue=ops.EPI_F32)
            teacher_done = ts.record_event()
        # _forward_student (:253-266): pooled feature -> Linear -> normalize
        f = self.s.forward(ws, x, want_pooled=True, want_tokens=False)
        sg_raw = ws.get("s.g", (B, Dt), torch.float32)
        ops.gemm(f["pooled"], P.b["head.weight"], sg_raw, M=B, N=Dt, K=Ds, epilogue=ops.EPI_F32, bias=P.p["head.bias"])
        sg = ws.get("s.gn", (B, Dt), torch.bfloat16)
        sinv = ws.get("s.inv", (B,), torch.float32)
        ops.l2norm_fwd(sg_raw, sg, sinv, B, Dt, 1e-12)
        main.wait_event(teacher_done)
        s_logits = ws.get("g.s_logits", (B, Qp), torch.float32)
        ops.gemm(sg, qb, s_logits, M=B, N=Qp, K=Dt, epilogue=ops.EPI_F32)
        dlg = ws.get("g.dlogits", (B, Qp), torch.bfloat16, zero=True)     # pad columns: zero, never written
        ops.kl_fwd_bwd(s_logits, t_logits, Qp, 1.0 / a.temperature, 1.0 / B, self._loss, dlg, Qp, B, Q)
        dsg_n = ws.get("g.dsg_n", (B, Dt), torch.float32)
        ops.gemm(dlg, qb, dsg_n, M=B, N=Dt, K=Qp, trans_b=True, epilogue=ops.EPI_F32)
        dsg = ws.get("g.dsg", (B, Dt), torch.bfloat16)
        ops.l2norm_bwd(dsg_n, sg_raw, sinv, dsg, B, Dt)
        self._head_wgrad("", dsg, f["pooled"], Dt, Ds, B)
        dpool = ws.get("s.dpool", (B, Ds), torch.float32)
        ops.gemm(dsg, P.b["head.weight"], dpool, M=B, N=Ds, K=Dt, trans_b=True, epilogue=ops.EPI_F32)
        self.s.backward(ws, f, B, dpool, None, self.side_stream)
        self._last = dict(lam=lam, index=index, t_logits=t_logits[:, :Q], s_logits=s_logits[:, :Q])
        return TrainingStepResult(loss=self._loss[0], log_dict={})

    def state_dict(self) -> Dict[str, Tensor]:
        out = self._backbone_state()
        out["student_projection_head.weight"] = self.student.p["head.weight"].detach().clone()
        out["student_projection_head.bias"] = self.student.p["head.bias"].detach().clone()
        out["teacher_queue"] = self.teacher_queue.detach().clone()
        return out


class DistillationV2(_DistillBase):
    def __init__(self, student_cfg: "ViTConfig | ResNetConfig", teacher_cfg: ViTConfig, method_args: Optional[DistillationV2Args] = None,
                 global_batch_size: int = 128, total_steps: int = 100_000, max_epochs: int = 100, device: "str | torch.device" = "cuda",
                 student_state: Optional[Dict[str, Tensor]] = None, teacher_state: Optional[Dict[str, Tensor]] = None,
                 head_state: Optional[Dict[str, Tensor]] = None, seed: int = 0) -> None:
        a = method_args or DistillationV2Args()
        Ds = student_cfg.feature_dim if isinstance(student_cfg, ResNetConfig) else student_cfg.embed_dim
        self.Dtt = a.n_teacher_blocks * teacher_cfg.embed_dim
        # DistillationV2Head (distillationv2.py:116-152): one Linear, or Linear-LayerNorm-GELU stacks of width `projection_hidden_dim`
        # closed by a Linear; Sequential indices 3i (Linear), 3i + 1 (LayerNorm)
        n, hid = max(a.n_projection_layers, 1), a.projection_hidden_dim
        if n > 1 and hid > 2048:
            raise NotImplementedError("projection_hidden_dim > 2048: the LayerNorm kernels hold a row in registers up to 2048 columns")
        self._lin: List[Tuple[str, int, int]] = []   # (name prefix, out, in) of the Linear layers
        self._ln: List[str] = []
        shapes: List[Tuple[str, Tuple[int, ...]]] = []
        d_in = Ds
        for i in range(n - 1):
            self._lin.append((f"mlp.{3 * i}.", hid, d_in))
            self._ln.append(f"mlp.{3 * i + 1}.")
            shapes += [(f"mlp.{3 * i}.weight", (hid, d_in)), (f"mlp.{3 * i}.bias", (hid,)), (f"mlp.{3 * i + 1}.weight", (hid,)), (f"mlp.{3 * i + 1}.bias", (hid,))]
            d_in = hid
        last = "mlp." if n == 1 else f"mlp.{3 * (n - 1)}."
        self._lin.append((last, self.Dtt, d_in))
        shapes += [(last + "weight", (self.Dtt, d_in)), (last + "bias", (self.Dtt,))]
        super().__init__(student_cfg, teacher_cfg, a, shapes, global_batch_size, total_steps, max_epochs, device, student_state, teacher_state,
                         head_state, seed)
        if head_state is None:      # nn.LayerNorm starts at weight 1 (the base class zero-fills 1-D head tensors: biases)
            for ln in self._ln:
                self.student.p["head." + ln + "weight"].fill_(1.0)
            self.student.bf16.copy_(self.student.data)
        self._tabs: Dict[Tuple[int, int, int, int], Any] = {}

    def _head_forward(self, x: Tensor, R: int) -> Tuple[Tensor, List[Dict[str, Tensor]]]:
        """tokens bf16 [R, Ds] -> projected fp32 [R, Dtt]; the saved activations of the hidden layers for the backward."""
        ws, P = self.ws, self.student
        saved: List[Dict[str, Tensor]] = []
        for i, ((lin, n_out, k_in), ln) in enumerate(zip(self._lin[:-1], self._ln)):
            y = ws.get(f"s.hy{i}", (R, n_out), torch.float32)
            ops.gemm(x, P.b["head." + lin + "weight"], y, M=R, N=n_out, K=k_in, epilogue=ops.EPI_F32, bias=P.p["head." + lin + "bias"])
            u = ws.get(f"s.hu{i}", (R, n_out), torch.bfloat16)
            mean, rstd = ws.get(f"s.hmean{i}", (R,), torch.float32), ws.get(f"s.hrstd{i}", (R,), torch.float32)
            ops.layernorm_fwd(y, P.p["head." + ln + "weight"], P.p["head." + ln + "bias"], R, n_out, y_bf16=u, mean=mean, rstd=rstd, eps=1e-5)
            h = ws.get(f"s.hh{i}", (R, n_out), torch.bfloat16, pad_rows=64)
            ops.gelu_fwd(u, h, R * n_out)
            saved.append(dict(x=x, y=y, u=u, mean=mean, rstd=rstd))
            x = h
        lin, n_out, k_in = self._lin[-1]
        s_proj = ws.get("s.proj", (R, n_out), torch.float32)
        ops.gemm(x, P.b["head." + lin + "weight"], s_proj, M=R, N=n_out, K=k_in, epilogue=ops.EPI_F32, bias=P.p["head." + lin + "bias"])
        saved.append(dict(x=x))
        return s_proj, saved

    def _head_backward(self, dsb: Tensor, saved: List[Dict[str, Tensor]], R: int) -> Tensor:
        """d(projected) bf16 [R, Dtt] -> d(tokens) fp32 [R, Ds]; parameter gradients accumulated."""
        ws, P = self.ws, self.student
        d = dsb
        for i in range(len(self._lin) - 1, -1, -1):
            lin, n_out, k_in = self._lin[i]
            self._head_wgrad(lin, d, saved[i]["x"], n_out, k_in, R)
            if i == 0:
                dtok = ws.get("s.dtok", (R, k_in), torch.float32)
                ops.gemm(d, P.b["head." + lin + "weight"], dtok, M=R, N=k_in, K=n_out, trans_b=True, epilogue=ops.EPI_F32)
                return dtok
            dh = ws.get(f"s.hdh{i}", (R, k_in), torch.bfloat16)
            ops.gemm(d, P.b["head." + lin + "weight"], dh, M=R, N=k_in, K=n_out, trans_b=True, epilogue=ops.EPI_BF16)
            sv, ln = saved[i - 1], self._ln[i - 1]
            ops.gelu_bwd(dh, sv["u"], dh, R * k_in)
            dy32 = ws.get(f"s.hdy32_{i}", (R, k_in), torch.float32)
            ops.layernorm_bwd(sv["y"], P.p["head." + ln + "weight"], sv["mean"], sv["rstd"], dh, None, dy32, P.g["head." + ln + "weight"],
                              P.g["head." + ln + "bias"], R, k_in)
            d = ws.get(f"s.hdy{i}", (R, k_in), torch.bfloat16, pad_rows=64)
            ops.cast_bf16(dy32, d)
        raise AssertionError("unreachable")

    def _resample(self, hs: int, ws_: int, ht: int, wt: int):
        key = (hs, ws_, ht, wt)
        if key not in self._tabs:
            (fi, fw, ft), (bi, bw, bt) = ops.resample_tables(hs, ws_, ht, wt, "bilinear")
            d = self.device
            self._tabs[key] = ((fi.to(d), fw.to(d), ft), (bi.to(d), bw.to(d), bt))
        return self._tabs[key]

    def training_step_impl(self, batch: Mapping[str, Any], batch_idx: int, mix: Optional[Tuple[float, Tensor]] = None) -> TrainingStepResult:
        a, ws, P = self.method_args, self.ws, self.student
        x, lam, index = self._mixup(batch, mix)
        B, Dt, Ds, Dtt = x.shape[0], self.tcfg.embed_dim, self.s.dim, self.Dtt
        main = torch.cuda.current_stream()
        if self._grad_sync is not None:
            self._grad_sync.reset()
        P.grad.zero_()
        self._loss.zero_()
        ts = self.teacher_stream
        ts.wait_event(main.record_event())
        with torch.cuda.stream(ts):   # _forward_teacher (distillationv2.py:224-257): last n blocks, normed, patch tokens, concatenated
            depth = self.tcfg.depth
            layers = list(range(depth - a.n_teacher_blocks, depth))
            tc = self.t_vit.forward(ws, "t", x, None, save=False, capture_layers=layers)
            Nt, pre_t = tc["N"], 1 + self.tcfg.num_register_tokens
            n_pt = Nt - pre_t
            r = torch.arange(B, dtype=torch.int64)
            prow = (r[:, None] * Nt + torch.arange(pre_t, Nt, dtype=torch.int64)[None, :]).reshape(-1).to(self.device)
            t_feat = ws.get("t.feat", (B * n_pt, Dtt), torch.float32)
            part = ws.get("t.part", (B * n_pt, Dt), torch.float32)
            for j, li in enumerate(layers):
                ops.gather_rows(tc["captured"][li].view(-1, Dt), Dt, prow, B * n_pt, Dt, out_f32=part)
                t_feat[:, j * Dt:(j + 1) * Dt].copy_(part)          # plumbing: channel concatenation
            teacher_done = ts.record_event()
        # _forward_student (:259-289): feature map -> head -> bilinear resize onto the teacher grid
        f = self.s.forward(ws, x, want_pooled=False, want_tokens=True)
        n_ps = f["n"]
        resize = (f["gh"], f["gw"]) != (tc["gh"], tc["gw"])
        s_proj, head_saved = self._head_forward(f["tokens"], B * n_ps)
        if resize:
            (fi, fw, ft), (bi, bw, bt) = self._resample(f["gh"], f["gw"], tc["gh"], tc["gw"])
            s_feat = ws.get("s.feat", (B * n_pt, Dtt), torch.float32)
            ops.resample_tokens(s_proj, fi, fw, s_feat, B, n_ps, n_pt, Dtt, ft)
        else:
            s_feat = s_proj
        main.wait_event(teacher_done)
        numel = B * n_pt * Dtt
        ds_feat = ws.get("s.dfeat32", (B * n_pt, Dtt), torch.float32)
        ops.mse_fwd_bwd(s_feat, t_feat, ds_feat, numel, 1.0 / numel, self._loss)
        if resize:
            ds_proj = ws.get("s.dproj32", (B * n_ps, Dtt), torch.float32)
            ops.resample_tokens(ds_feat, bi, bw, ds_proj, B, n_pt, n_ps, Dtt, bt)
        else:
            ds_proj = ds_feat
        dsb = ws.get("s.dproj", (B * n_ps, Dtt), torch.bfloat16)
        ops.cast_bf16(ds_proj, dsb)
        dtok = self._head_backward(dsb, head_saved, B * n_ps)
        self.s.backward(ws, f, B, None, dtok, self.side_stream)
        self._last = dict(lam=lam, index=index, t_feat=t_feat, s_feat=s_feat)
        return TrainingStepResult(loss=self._loss[0], log_dict={})

    def state_dict(self) -> Dict[str, Tensor]:
        out = self._backbone_state()
        for n in self.student.names:
            if n.startswith("head."):
                out["student_projection_head." + n[5:]] = self.student.p[n].detach().clone()
        return out
