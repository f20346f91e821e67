"""DistillationV3 on MI355X: frozen ViT teacher -> ViT or ResNet student, the whole step in HIP.

Mirrors LT/_methods/distillationv3/distillationv3.py: `DistillationV3Args` (:109-169), `DistillationV3.training_step_impl`
(:235-273), `_mixup_data` (:356-368), `_forward_teacher` (:293-322), `_forward_student` (:324-354), `_update_queue` (:275-291),
`DistillationV3Loss.forward` (distillationv3_loss.py:35-117), `configure_gradient_clipping` (:400-410, norm 1.0) and the generic
`Method.configure_optimizers` (LT/_methods/method.py:89-121: AdamW, sqrt LR scaling, CosineWarmupScheduler).

Teacher: DINOv3 ViT (RoPE, storage tokens; `dinov3.py`) or a DINOv2 ViT on `vit.ViTEngine`; student: a DINOv2 / DINOv3 ViT on the
same engine, or the torchvision ResNet-50 of BASELINE configs[3] on `resnet.ResNetEngine` (NHWC convolutions as MFMA GEMMs,
training-mode BatchNorm): its layer4 map [B, 2048, 7, 7] is the token matrix [B*49, 2048], the pooled feature its average
(`ResNetModelWrapper.forward_pool`, LT/_models/torchvision/resnet.py:40-44), the local projection is resized 7x7 -> 14x14.
State-dict names follow the reference: `student_embedding_model.wrapped_model._model.*` (ViT) /
`student_embedding_model.wrapped_model._features.*` (ResNet: the wrapper's IntermediateLayerGetter), `student_projection_head_global.*`,
`student_projection_head_local.*`, `teacher_queue`.

Optimizers: AdamW (the method's "auto") and `optimizer="lars"` (DistillationV3LARSArgs) on `lars.FlatLARS`.
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass
from typing import Any, Dict, List, Mapping, Optional, Tuple

import torch
import torch.distributed as dist
from torch import Tensor

from . import ops
from .parallel import GradSync
from .lars import FlatLARS, LARSArgs
from .params import FlatParams
from .schedules import warmup_cosine_lr_factor
from .resnet import ResNetConfig, ResNetEngine, flat_named, init_resnet_state
from .vit import ViTConfig, ViTEngine, Workspace, _split_k, split_k_plan, vit_param_shapes

NO_DECAY_KEYS = ("cls_token", "mask_token", "storage_token", "register_token", "pos_embed")


@dataclass
class DistillationV3Args:
    """Method + AdamW arguments (distillationv3.py:109-193).  weight_decay None = the reference's "auto": 0.04 for transformer
    students, 1e-6 for convolutional ones (DistillationV3AdamWArgs.resolve_auto, :163-170)."""
    queue_size: int = 8192
    temperature_global: float = 0.07
    temperature_local: float = 0.07
    lr_scale_method: str = "sqrt"
    reference_batch_size: int = 1536
    loss_local_weight: float = 1.0
    lr: float = 0.0005
    betas: Tuple[float, float] = (0.9, 0.999)
    eps: float = 1e-8
    weight_decay: Optional[float] = None
    gradient_clip_val: float = 1.0
    optimizer: str = "adamw"          # the method's "auto" (distillationv3.py:380-388); "lars": DistillationV3LARSArgs (:147-157)
    lars: Optional[LARSArgs] = None   # None = DistillationV3LARSArgs' values


@dataclass
class TrainingStepResult:
    loss: Tensor
    log_dict: Mapping[str, Any]


def weight_decays(name: str, shape: torch.Size) -> bool:
    """LT/_optim/optimizer_helpers.py:83-175 restricted to the modules on this path: no decay for norm layers, biases,
    <= 1-D parameters (LayerScale), tokens and positional embeddings."""
    if len(shape) <= 1 or "norm" in name or name.endswith("bias") or any(k in name for k in NO_DECAY_KEYS):
        return False
    return True


class _Trainer:
    def __init__(self, total_steps: int, max_epochs: int) -> None:
        self.global_step = 0
        self.max_epochs = max_epochs
        self.estimated_stepping_batches = total_steps


class DistillationV3:
    def __init__(self, student_cfg: "ViTConfig | ResNetConfig", teacher_cfg: ViTConfig, method_args: Optional[DistillationV3Args] = None,
                 global_batch_size: int = 128, total_steps: int = 100_000, max_epochs: int = 100, device: str | torch.device = "cuda",
                 student_state: Optional[Dict[str, Tensor]] = None, teacher_state: Optional[Dict[str, Tensor]] = None,
                 proj_global_state: Optional[Dict[str, Tensor]] = None, proj_local_state: Optional[Dict[str, Tensor]] = None,
                 seed: int = 0) -> None:
        from .vit import init_vit_state

        self.method_args = a = method_args or DistillationV3Args()
        self.scfg, self.tcfg = student_cfg, teacher_cfg
        self.conv_student = isinstance(student_cfg, ResNetConfig)
        if a.weight_decay is None:
            a.weight_decay = 1e-6 if self.conv_student else 0.04
        self.device = dev = torch.device(device)
        ops.require_device(dev, "DistillationV3")
        g = torch.Generator().manual_seed(seed)
        Ds, Dt = (student_cfg.feature_dim if self.conv_student else student_cfg.embed_dim), teacher_cfg.embed_dim
        self.Ds = Ds
        def random_init(c: ViTConfig) -> Dict[str, Tensor]:
            sd = init_vit_state(c, g)
            if c.rope_base is not None:          # RoPE models have no positional table: the engine's slot stays zero
                sd["pos_embed"] = torch.zeros_like(sd["pos_embed"])
            if c.mask_k_bias:
                for k in sd:
                    if k.endswith("attn.qkv.bias"):
                        sd[k][c.embed_dim:2 * c.embed_dim] = 0
            return sd

        if self.conv_student:
            sb = student_state if student_state is not None else init_resnet_state(student_cfg, g)
        else:
            sb = student_state if student_state is not None else random_init(student_cfg)
        tb = teacher_state if teacher_state is not None else random_init(teacher_cfg)

        def lin(state: Optional[Dict[str, Tensor]]) -> Dict[str, Tensor]:
            if state is not None:
                return {k: v.detach().clone().float() for k, v in state.items()}
            w = torch.nn.init.trunc_normal_(torch.empty(Dt, Ds), std=0.02, generator=g)   # distillationv3.py:212-213
            bound = 1.0 / math.sqrt(Ds)
            return {"weight": w, "bias": torch.empty(Dt).uniform_(-bound, bound, generator=g)}

        pg, pl = lin(proj_global_state), lin(proj_local_state)
        if self.conv_student:
            named: List[Tuple[str, Tensor]] = flat_named(student_cfg, sb, "backbone.")
        else:
            named = [("backbone." + n, sb[n]) for n, _ in vit_param_shapes(student_cfg)]
        named += [("proj_global.weight", pg["weight"]), ("proj_global.bias", pg["bias"]),
                  ("proj_local.weight", pl["weight"]), ("proj_local.bias", pl["bias"])]
        self.student = FlatParams(named, dev, True)
        self.teacher = FlatParams([(n, tb[n]) for n, _ in vit_param_shapes(teacher_cfg)], dev, False)
        if self.conv_student:
            self.s_net = ResNetEngine(student_cfg, self.student, "backbone.", buffers=sb)
            self.s_vit = None
            # the classifier is part of the exported torchvision state_dict but not of the wrapper's trained parameters
            self._fc = {k: sb[k].detach().clone() for k in ("fc.weight", "fc.bias") if k in sb}
        else:
            self.s_vit = ViTEngine(student_cfg, self.student, "backbone.")
        self.t_vit = ViTEngine(teacher_cfg, self.teacher, "")
        self.ws = Workspace(dev)
        self.teacher_queue = torch.zeros(a.queue_size, Dt, device=dev)
        self.global_batch_size = global_batch_size
        self.trainer = _Trainer(total_steps, max_epochs)
        # ---- optimizer (method.py:89-121)
        scale = global_batch_size / a.reference_batch_size
        if a.lr_scale_method == "sqrt":
            scale = math.sqrt(scale)
        if a.optimizer not in ("adamw", "lars"):
            raise ValueError(f"Invalid optimizer type: '{a.optimizer}'")
        lars_args = (a.lars or LARSArgs()) if a.optimizer == "lars" else None
        self.base_lr = (lars_args.lr if lars_args is not None else a.lr) * scale
        warm_epochs = min(10, max_epochs / 10)
        self.warmup_steps = min(int(total_steps), int(total_steps / max(1, max_epochs) * warm_epochs))
        self.lars = FlatLARS(self.student, lars_args) if lars_args is not None else None
        self.exp_avg = torch.zeros_like(self.student.data) if self.lars is None else None
        self.exp_avg_sq = torch.zeros_like(self.student.data) if self.lars is None else None
        self.seg_lr = torch.full((len(self.student.names),), self.base_lr, dtype=torch.float32, device=dev)
        self.seg_wd_on = torch.tensor([1 if weight_decays(n, self.student.shapes[n]) else 0 for n in self.student.names],
                                      dtype=torch.uint8, device=dev)
        self.seg_frozen = torch.zeros(len(self.student.names), dtype=torch.uint8, device=dev)
        self._sumsq = torch.zeros(1, device=dev)
        self._loss_slots = torch.zeros(2, device=dev)
        self.opt_step = 0
        self.last_grad_norm: Optional[Tensor] = None
        self._grad_sync: Optional[GradSync] = None
        self.teacher_stream = torch.cuda.Stream(device=dev)
        self.side_stream = torch.cuda.Stream(device=dev)
        self.reduce_stream = torch.cuda.Stream(device=dev)
        # data parallel: per-block gradient all-reduces issued during backward (see dinov2.py); LT_GRAD_OVERLAP=0 = after it
        self.overlap_grad_reduce = os.environ.get("LT_GRAD_OVERLAP", "1") != "0"
        self._proj_span = self.student.span(("proj_global.", "proj_local."))
        self._block_spans = [] if self.conv_student else [self.student.span((f"backbone.blocks.{i}.",)) for i in range(student_cfg.depth)]
        self._idx: Dict[Tuple[int, int, int], Tuple[Tensor, Tensor]] = {}
        self._resample_tabs: Dict[Tuple[int, int, int, int], Any] = {}

    # ------------------------------------------------------------------ helpers
    @property
    def world(self) -> int:
        return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1

    def _rows(self, B: int, N: int, prefix: int) -> Tuple[Tensor, Tensor]:
        """Row indices of the cls tokens and of the patch tokens in a [B*N, D] token matrix."""
        key = (B, N, prefix)
        if key not in self._idx:
            r = torch.arange(B, dtype=torch.int64)
            cls = r * N
            patch = (r[:, None] * N + torch.arange(prefix, N, dtype=torch.int64)[None, :]).reshape(-1)
            self._idx[key] = (cls.to(self.device), patch.to(self.device))
        return self._idx[key]

    def _resample(self, hs: int, ws_: int, ht: int, wt: int):
        key = (hs, ws_, ht, wt)
        if key not in self._resample_tabs:
            (fi, fw, ft), (bi, bw, bt) = ops.resample_tables(hs, ws_, ht, wt, "bilinear")
            dev = self.device
            self._resample_tabs[key] = ((fi.to(dev), fw.to(dev), ft), (bi.to(dev), bw.to(dev), bt))
        return self._resample_tabs[key]

    def _normalize(self, tag: str, x: Tensor, rows: int, D: int) -> Tuple[Tensor, Tensor]:
        """F.normalize(x, dim=-1): returns (bf16 normalised rows, 1/||x||)."""
        y = self.ws.get(tag + ".n", (rows, D), torch.bfloat16)
        inv = self.ws.get(tag + ".inv", (rows,), torch.float32)
        ops.l2norm_fwd(x, y, inv, rows, D, 1e-12)
        return y, inv

    # ------------------------------------------------------------------ the step
    def training_step_impl(self, batch: Dict[str, Any], batch_idx: int, mix: Optional[Tuple[float, Tensor]] = None) -> TrainingStepResult:
        a, ws, dev = self.method_args, self.ws, self.device
        Ds, Dt = self.Ds, self.tcfg.embed_dim
        views = batch["views"][0].to(dev, torch.float32, non_blocking=True).contiguous()
        B = views.shape[0]
        # ---- mixup (:356-368): lambda ~ U(0,1), random permutation -- same host RNG draws, same order, as the reference
        if mix is None:
            lam = torch.empty(1).uniform_(0.0, 1.0).item()
            index = torch.randperm(B)
        else:
            lam, index = mix
        x = ws.get("mix.x", tuple(views.shape), torch.float32)
        ops.mixup(views, index.to(dev, torch.int64), float(lam), x)

        main = torch.cuda.current_stream()
        if self._grad_sync is not None:
            self._grad_sync.reset()
        self.student.grad.zero_()
        self._loss_slots.zero_()
        # ---- teacher (no grad) on its own stream
        ts = self.teacher_stream
        ts.wait_event(main.record_event())
        with torch.cuda.stream(ts):
            tc = self.t_vit.forward(ws, "t", x, None, save=False)
            Nt, pre_t = tc["N"], 1 + self.tcfg.num_register_tokens
            n_pt = Nt - pre_t
            t_cls_rows, t_patch_rows = self._rows(B, Nt, pre_t)
            txn = tc["xn"].view(-1, Dt)
            tg_raw = ws.get("t.g", (B, Dt), torch.float32)
            tl_raw = ws.get("t.l", (B * n_pt + 8, Dt), torch.float32, zero=True)   # (+8 zero rows: padded-K reads of the batched GEMMs)
            ops.gather_rows(txn, Dt, t_cls_rows, B, Dt, out_f32=tg_raw)
            ops.gather_rows(txn, Dt, t_patch_rows, B * n_pt, Dt, out_f32=tl_raw)
            tg, tg_inv = self._normalize("t.g", tg_raw, B, Dt)
            tl, _ = self._normalize("t.l", tl_raw, B * n_pt + 8, Dt)
            # queue update (:275-291) with the fp32 normalised global features
            tgn = ws.get("t.gn", (B, Dt), torch.float32)
            torch.mul(tg_raw, tg_inv[:, None], out=tgn)          # plumbing: B x Dt scale for the fp32 queue rows
            Q = self.teacher_queue.shape[0]
            if B >= Q:
                self.teacher_queue.copy_(tgn[:Q])
            else:
                self.teacher_queue[B:] = self.teacher_queue[:-B].clone()
                self.teacher_queue[:B] = tgn
            # [B, Q] similarity rows are Qp = Q rounded up to 8 columns apart (16-byte rows for the GEMMs); pad rows of the bf16 queue copy
            # stay zero, pad columns of the logits are never read (lt_kl_fwd_bwd takes the row stride)
            Qp = (Q + 7) // 8 * 8
            qb = ws.get("queue.bf16", (Qp, Dt), torch.bfloat16, zero=True)
            ops.cast_bf16(self.teacher_queue, qb[:Q])
            t_logits = ws.get("g.t_logits", (B, Qp), torch.float32)
            ops.gemm(tg, qb, t_logits, M=B, N=Qp, K=Dt, epilogue=ops.EPI_F32)
            n_pad = (n_pt + 7) // 8 * 8
            St = ws.get("l.St", (B * n_pt, n_pad), torch.float32)
            ops.gemm(tl, tl, St, M=n_pt, N=n_pt, K=Dt, epilogue=ops.EPI_F32, ldc=n_pad, batch=B, stride_a=n_pt * Dt, stride_b=n_pt * Dt,
                     stride_c=n_pt * n_pad)
            teacher_done = ts.record_event()

        # ---- student forward
        s_rope = None
        if self.conv_student:
            # ResNet student (distillationv3.py:324-354 with ResNetModelWrapper): layer4 map as tokens, pooled = their average
            sc = self.s_net.forward(ws, "s", x, save=True, train=True)
            Ds = self.Ds
            n_ps = sc["h"] * sc["w"]
            sc["gh"], sc["gw"] = sc["h"], sc["w"]
            resize = (sc["gh"], sc["gw"]) != (tc["gh"], tc["gw"])
            sl_in = sc["feat"]                                                    # bf16 [B * n_ps (+pad), Ds]
            sg_in = ws.get("s.g_in", (B, Ds), torch.bfloat16)
            ops.token_mean(sl_in, sg_in, B, n_ps, Ds)
        else:
            if self.scfg.rope_base is not None:   # DINOv3 student in training mode: per-block RoPE tables, drawn after the mixup draws
                p_ = self.scfg.patch_size
                s_rope = self.s_vit.rope_tables_train(-(-x.shape[2] // p_), -(-x.shape[3] // p_))
            sc = self.s_vit.forward(ws, "s", x, None, save=True, rope_tables=s_rope)
            Ns, pre_s = sc["N"], 1 + self.scfg.num_register_tokens
            n_ps = Ns - pre_s
            resize = (sc["gh"], sc["gw"]) != (tc["gh"], tc["gw"])   # bilinear resize of the student map onto the teacher grid (:338-345)
            s_cls_rows, s_patch_rows = self._rows(B, Ns, pre_s)
            sxn = sc["xn"].view(-1, Ds)
            sg_in = ws.get("s.g_in", (B, Ds), torch.bfloat16)
            sl_in = ws.get("s.l_in", (B * n_ps, Ds), torch.bfloat16)
            ops.gather_rows(sxn, Ds, s_cls_rows, B, Ds, out_bf16=sg_in)
            ops.gather_rows(sxn, Ds, s_patch_rows, B * n_ps, Ds, out_bf16=sl_in)
        P = self.student
        n_pl = n_pt if resize else n_ps            # tokens per image of the (resized) student local features
        sg_raw = ws.get("s.g", (B, Dt), torch.float32)
        sl_raw = ws.get("s.l", (B * n_pl + 8, Dt), torch.float32, zero=True)
        ops.gemm(sg_in, P.b["proj_global.weight"], sg_raw, M=B, N=Dt, K=Ds, epilogue=ops.EPI_F32, bias=P.p["proj_global.bias"])
        if resize:
            tabs = self._resample(sc["gh"], sc["gw"], tc["gh"], tc["gw"])
            sl_proj = ws.get("s.l_proj", (B * n_ps, Dt), torch.float32)
            ops.gemm(sl_in, P.b["proj_local.weight"], sl_proj, M=B * n_ps, N=Dt, K=Ds, epilogue=ops.EPI_F32, bias=P.p["proj_local.bias"])
            (fi, fw, ft), _ = tabs
            ops.resample_tokens(sl_proj, fi, fw, sl_raw, B, n_ps, n_pt, Dt, ft)
        else:
            ops.gemm(sl_in, P.b["proj_local.weight"], sl_raw, M=B * n_ps, N=Dt, K=Ds, epilogue=ops.EPI_F32, bias=P.p["proj_local.bias"])
        sg, sg_inv = self._normalize("s.g", sg_raw, B, Dt)
        sl, sl_inv = self._normalize("s.l", sl_raw, B * n_pl + 8, Dt)

        # ---- losses (distillationv3_loss.py:60-115) and their gradients w.r.t. the similarity logits
        main.wait_event(teacher_done)
        Q = self.teacher_queue.shape[0]
        s_logits = ws.get("g.s_logits", (B, Qp), torch.float32)
        ops.gemm(sg, qb, s_logits, M=B, N=Qp, K=Dt, epilogue=ops.EPI_F32)
        dlg = ws.get("g.dlogits", (B, Qp), torch.bfloat16, zero=True)     # pad columns: zero, never written
        ops.kl_fwd_bwd(s_logits, t_logits, Qp, 1.0 / a.temperature_global, 1.0 / B, self._loss_slots[0:], dlg, Qp, B, Q)
        Ss = ws.get("l.Ss", (B * n_pl, n_pad), torch.float32)
        ops.gemm(sl, sl, Ss, M=n_pl, N=n_pl, K=Dt, epilogue=ops.EPI_F32, ldc=n_pad, batch=B, stride_a=n_pl * Dt, stride_b=n_pl * Dt,
                 stride_c=n_pl * n_pad)
        dS = ws.get("l.dS", (B * n_pl, n_pad), torch.bfloat16, zero=True)     # pad columns stay zero
        ops.kl_fwd_bwd(Ss, St, n_pad, 1.0 / a.temperature_local, a.loss_local_weight / (B * n_pl), self._loss_slots[1:], dS, n_pad,
                       B * n_pl, n_pl)

        # ---- backward: similarity logits -> normalised features -> projection heads -> student tokens
        dsg_n = ws.get("g.dsg_n", (B, Dt), torch.float32)
        ops.gemm(dlg, qb, dsg_n, M=B, N=Dt, K=Qp, trans_b=True, epilogue=ops.EPI_F32)
        G = ws.get("l.G", (B * n_pl, n_pad), torch.bfloat16, zero=True)
        ops.symmetrize_bf16(dS, G, B, n_pl, n_pad)
        dsl_n = ws.get("l.dsl_n", (B * n_pl, Dt), torch.float32)
        ops.gemm(G, sl, dsl_n, M=n_pl, N=Dt, K=n_pad, trans_b=True, epilogue=ops.EPI_F32, lda=n_pad, batch=B, stride_a=n_pl * n_pad,
                 stride_b=n_pl * Dt, stride_c=n_pl * Dt)
        dsg = ws.get("g.dsg", (B, Dt), torch.bfloat16)
        dsl = ws.get("l.dsl", (B * n_ps, Dt), torch.bfloat16)
        ops.l2norm_bwd(dsg_n, sg_raw, sg_inv, dsg, B, Dt)
        if resize:   # gradient of the resized features -> transposed bilinear map -> projection output
            dsl_r = ws.get("l.dsl_r", (B * n_pl, Dt), torch.bfloat16)
            ops.l2norm_bwd(dsl_n, sl_raw, sl_inv, dsl_r, B * n_pl, Dt)
            dsl_rf = ws.get("l.dsl_rf", (B * n_pl, Dt), torch.float32)
            dsl_rf.copy_(dsl_r)                                   # plumbing: dtype of the resampling kernel's input
            dsl_pf = ws.get("l.dsl_pf", (B * n_ps, Dt), torch.float32)
            _, (bi, bw, bt) = tabs
            ops.resample_tokens(dsl_rf, bi, bw, dsl_pf, B, n_pl, n_ps, Dt, bt)
            ops.cast_bf16(dsl_pf, dsl)
        else:
            ops.l2norm_bwd(dsl_n, sl_raw, sl_inv, dsl, B * n_ps, Dt)
        slab = ws.get("wgrad.slabs", (32 * 1024 * 1024,), torch.float32)
        for tagp, dy, xin, rows in (("proj_global", dsg, sg_in, B), ("proj_local", dsl, sl_in, B * n_ps)):
            ops.colsum_bf16(dy, P.g[tagp + ".bias"], rows, Dt)
            tiles = ((Dt + 127) // 128) * ((Ds + 127) // 128)   # few output tiles, long contraction: split-K into slabs
            ops.gemm(dy, xin, P.g[tagp + ".weight"], M=Dt, N=Ds, K=rows, trans_a=True, trans_b=True, epilogue=ops.EPI_F32_ACCUM, lda=Dt, ldb=Ds,
                     workspace=slab, **split_k_plan(Dt, Ds, rows, True, _split_k(tiles, rows)))
        dcls = ws.get("s.dcls", (B, Ds), torch.float32)
        dpat = ws.get("s.dpat", (B * n_ps, Ds), torch.float32)
        ops.gemm(dsg, P.b["proj_global.weight"], dcls, M=B, N=Ds, K=Dt, trans_b=True, epilogue=ops.EPI_F32)
        ops.gemm(dsl, P.b["proj_local.weight"], dpat, M=B * n_ps, N=Ds, K=Dt, trans_b=True, epilogue=ops.EPI_F32)
        sync = self._gradient_sync() if self.overlap_grad_reduce else None
        done_blocks: List[int] = []
        if sync is not None:
            sync.start(*self._proj_span)   # projection heads are final: reduce them under the backbone backward
        if self.conv_student:
            # gradient reaching the layer4 map: the local path per position + the pooled path spread over the n positions
            dfeat = ws.get("s.dfeat", (sc["feat"].shape[0], Ds), torch.bfloat16, zero=True)
            ops.pool_bwd_add(dpat, dcls, dfeat, B, n_ps, Ds)
            self.s_net.backward(ws, sc, dfeat, side=self.side_stream)
            main.wait_stream(self.side_stream)
        else:
            dxn = ws.get("s.dxn", (B * Ns, Ds), torch.float32)
            dxn.zero_()
            ops.scatter_add_rows(dcls, s_cls_rows, dxn, Ds, B, Ds)
            ops.scatter_add_rows(dpat, s_patch_rows, dxn, Ds, B * n_ps, Ds)
            blk = self.scfg.depth
            for ev in self.s_vit.backward_iter(ws, sc, dxn.view(B, Ns, Ds), side=self.side_stream):
                if ev == "block":
                    blk -= 1
                    if sync is not None:   # block `blk` is final: LayerScale gradients, then its all-reduce, beside the rest of backward
                        rs = self.reduce_stream
                        rs.wait_event(main.record_event())
                        rs.wait_event(self.side_stream.record_event())
                        with torch.cuda.stream(rs):
                            self.s_vit.finish_layerscale_grads(blocks=[blk], last_call=False)
                            sync.start(*self._block_spans[blk])
                        done_blocks.append(blk)
            main.wait_stream(self.side_stream)
            if sync is not None:
                main.wait_stream(self.reduce_stream)
            self.s_vit.finish_layerscale_grads(blocks=[i for i in range(self.scfg.depth) if i not in done_blocks])

        ls = self._loss_slots
        w = a.loss_local_weight
        logs = {"train_loss/global_loss": ls[0], "train_loss/local_loss": ls[1] / w if w else ls[1]}
        self._last = dict(B=B, lam=lam, index=index, t_logits=t_logits[:, :Q], s_logits=s_logits[:, :Q], tg=tg, tl=tl, sg=sg, sl=sl)
        return TrainingStepResult(loss=ls[0] + ls[1], log_dict=logs)

    def synced_logs(self, res: "TrainingStepResult") -> Dict[str, Tensor]:
        """`train_loss` + `log_dict` averaged over ranks in one coalesced all-reduce (what `Method.training_step` logs with
        `sync_dist=True`, method.py:131-144)."""
        from .parallel import coalesced_mean

        keys = ["train_loss"] + list(res.log_dict)
        vals = coalesced_mean([res.loss] + [torch.as_tensor(res.log_dict[k], device=res.loss.device) for k in res.log_dict])
        return dict(zip(keys, vals))

    # ------------------------------------------------------------------ optimizer hooks
    def _gradient_sync(self) -> Optional[GradSync]:
        if self.world == 1:
            return None
        if self._grad_sync is None:
            self._grad_sync = GradSync(self.student.grad)
        return self._grad_sync

    def optimizer_step(self) -> None:
        a = self.method_args
        k = self.trainer.global_step
        total = int(self.trainer.estimated_stepping_batches)
        lr_factor = warmup_cosine_lr_factor(k, self.warmup_steps, total, 0.001)   # CosineWarmupScheduler default end_value
        sync = self._gradient_sync()
        if sync is not None:
            sync.finish()   # what backward has not started yet (embeddings, final norm), wait, 1/world
        self._sumsq.zero_()
        ops.sumsq(self.student.grad, self._sumsq)
        self.opt_step += 1
        if self.lars is not None:
            self.lars.step(self.seg_lr, self.seg_wd_on, lr_factor, self._sumsq, a.gradient_clip_val)
        else:
            ops.adamw_flat(self.student.data, self.student.grad, self.exp_avg, self.exp_avg_sq, self.student.bf16, self.student.seg_of_chunk,
                           self.seg_lr, self.seg_wd_on, self.seg_frozen, False, lr_factor, a.weight_decay, a.betas[0], a.betas[1], a.eps,
                           self.opt_step, self._sumsq, a.gradient_clip_val)
        (self.s_net if self.conv_student else self.s_vit).refresh_padded_weights()
        self.last_grad_norm = self._sumsq
        self.trainer.global_step += 1

    def train_step(self, views: Tensor, mix: Optional[Tuple[float, Tensor]] = None) -> TrainingStepResult:
        res = self.training_step_impl({"views": [views]}, 0, mix=mix)
        self.optimizer_step()
        return res

    def export_backbone_state_dict(self) -> Dict[str, Tensor]:
        """What the reference exports after distillation: `get_model().state_dict()` of the STUDENT (the torchvision ResNet with
        its untouched classifier, or the ViT)."""
        if self.conv_student:
            return self.s_net.state_dict(extra=self._fc)
        bb = {n[9:]: self.student.p[n].detach().clone() for n in self.student.names if n.startswith("backbone.")}
        if self.scfg.rope_base is not None:
            from .dinov3 import export_dinov3_state

            bb = export_dinov3_state(bb, self.scfg)
        return bb

    def state_dict(self) -> Dict[str, Tensor]:
        """Reference key names; the teacher is left out like `on_save_checkpoint` does (:415-423)."""
        out: Dict[str, Tensor] = {}
        if self.conv_student:   # ResNetModelWrapper registers `_features` (conv1 .. layer4) and `_pool`; `fc` is not a submodule
            for k, v in self.s_net.state_dict().items():
                out["student_embedding_model.wrapped_model._features." + k] = v
            bb = {}
        else:
            bb = {n[9:]: self.student.p[n].detach().clone() for n in self.student.names if n.startswith("backbone.")}
        if not self.conv_student and self.scfg.rope_base is not None:
            from .dinov3 import export_dinov3_state

            bb = export_dinov3_state(bb, self.scfg)
        if not self.conv_student and self.scfg.rope_base is None:
            from .checkpoint import vit_key_from_flat

            bb = {vit_key_from_flat(k, self.scfg.depth, self.scfg.block_chunks): v for k, v in bb.items()}   # chunked DINOv2 students
        for k, v in bb.items():
            out["student_embedding_model.wrapped_model._model." + k] = v
        for n in self.student.names:
            v = self.student.p[n].detach().clone()
            if n.startswith("backbone."):
                continue
            elif n.startswith("proj_global."):
                out["student_projection_head_global." + n[12:]] = v
            else:
                out["student_projection_head_local." + n[11:]] = v
        out["teacher_queue"] = self.teacher_queue.detach().clone()
        return out

    def load_state_dict(self, sd: Mapping[str, Tensor], strict: bool = True) -> None:
        """Load what `state_dict()` (or the reference's DistillationV3.state_dict()) wrote: student backbone, both projection heads, queue."""
        from . import checkpoint
        checkpoint.distill_load_state_dict(self, sd, {"student_projection_head_global.": "proj_global.", "student_projection_head_local.": "proj_local."}, strict)

    def optimizer_state(self) -> Dict[str, Any]:
        """Optimizer moments (AdamW) / momentum buffer (LARS) and the step counters, for an exact resume together with `state_dict()`."""
        from . import checkpoint
        return checkpoint.distill_optimizer_state(self)

    def load_optimizer_state(self, st: Mapping[str, Any]) -> None:
        from . import checkpoint
        checkpoint.distill_load_optimizer_state(self, st)
