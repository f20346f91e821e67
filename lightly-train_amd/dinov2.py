"""DINOv2 method on the MI355X kernels -- mirrors lightly_train's `DINOv2(Method)`
(LT/_methods/dinov2/dinov2.py:179-660): same hyper-parameters (DINOv2Args), same step semantics
(training_step_impl -> on_before_optimizer_step -> clip -> AdamW -> LR schedule -> EMA), same parameter /
state_dict names, but the forward-backward underneath is hand-written HIP (no autograd, no Lightning).

Differences that are part of the design (all documented in DESIGN.md):
  * `training_step_impl` runs forward AND backward: gradients are written straight into the flat grad
    buffer ("a whole-step fwd+bwd call that writes .grad directly"); the returned loss is a
    detached device scalar.
  * student global+patch+local head rows go through the projection head as ONE batch of rows (the head is
    shared, dinov2.py:230-234), teacher cls+patch rows likewise.
"""
from __future__ import annotations

import os

import math
from dataclasses import dataclass, field
from typing import Any, Dict, List, Literal, Mapping, Optional, Sequence, Tuple

import torch
import torch.distributed as dist
from torch import Tensor

from . import checkpoint, ops
from .masking import MaskingGenerator, MaskProducer, create_collated_masks
from .parallel import GradSync
from .params import FlatParams
from .schedules import cosine_schedule, linear_warmup_schedule, warmup_cosine_lr_factor
from .vit import JointWgrad, ViTConfig, ViTEngine, Workspace, _split_k, init_vit_state, make_drop_plan, padded_rows, split_k_plan, vit_param_shapes


@dataclass
class DINOv2Args:
    """Field-for-field the reference's DINOv2Args (dinov2.py:70-153) + DINOv2AdamWViTArgs (:156-164)."""
    ibot_separate_head: bool = False
    hidden_dim: int = 2048
    dino_bottleneck_dim: int = 256
    ibot_bottleneck_dim: int = 256
    output_dim: int = 65536
    batch_norm: bool = False
    student_freeze_last_layer_steps: int = 1250
    student_freeze_backbone_steps: int = 0
    dino_loss_weight: float = 1.0
    ibot_loss_weight: float = 1.0
    koleo_loss_weight: float = 0.1
    center_method: Literal["softmax", "sinkhorn_knopp"] = "softmax"
    # Not a reference argument: hold the prototype logits of both heads in bf16 -- what the reference's own bf16-mixed run holds (the
    # prototype Linear runs under autocast, dinov2_head.py:66-71; the losses cast back with .float(), dinov2_loss.py:37-38,88) -- instead of
    # fp32 (wider than the reference's).  Halves the bytes of the serial logit section of the step; applies to the fused softmax-centering path
    # (center_method="softmax"), the other paths keep fp32.  LT_BF16_LOGITS=1 / 0 overrides.  Measured: profiles/r06_bf16_logits.md.
    bf16_logits: bool = False
    center_momentum: float = 0.9
    momentum_start: float = 0.992
    momentum_end: float = 1.0
    student_temp: float = 0.1
    teacher_temp_start: float = 0.04
    teacher_temp_end: float = 0.07
    teacher_temp_warmup_steps: int = 37500
    mask_ratio_min: float = 0.1
    mask_ratio_max: float = 0.5
    mask_probability: float = 0.5
    min_lr: float = 1.0e-06
    warmup_steps: int = 12500
    layerwise_decay: float = 0.9
    patch_embed_lr_multiplier: float = 0.2
    lr_scale_method: Literal["linear", "sqrt"] = "sqrt"
    reference_batch_size: int = 1024
    weight_decay_start: float = 0.04
    weight_decay_end: float = 0.4
    gradient_clip_val: float = 3.0
    # optimizer (DINOv2AdamWViTArgs)
    lr: float = 0.004
    betas: Tuple[float, float] = (0.9, 0.999)
    eps: float = 1e-8


@dataclass
class TrainingStepResult:
    loss: Tensor
    log_dict: Optional[Mapping[str, Any]] = None


WN_G, WN_V = "last_layer.parametrizations.weight.original0", "last_layer.parametrizations.weight.original1"
BN_BUFFERS = ("running_mean", "running_var", "num_batches_tracked")


def head_layer_names(use_bn: bool) -> Tuple[Tuple[str, str, str], Tuple[str, ...]]:
    """Sequential indices of _build_mlp (dinov2_head.py:74-99): Linear, [BatchNorm1d,] GELU, Linear, [BatchNorm1d,] GELU, Linear."""
    return (("mlp.0", "mlp.3", "mlp.6"), ("mlp.1", "mlp.4")) if use_bn else (("mlp.0", "mlp.2", "mlp.4"), ())


def head_param_shapes(in_dim: int, hidden: int, bottleneck: int, out_dim: int, use_bn: bool = False) -> List[Tuple[str, Tuple[int, ...]]]:
    """named_parameters() of one DINOv2ProjectionHead, in the reference's registration order."""
    lin, bnl = head_layer_names(use_bn)
    out: List[Tuple[str, Tuple[int, ...]]] = []
    for i, (l, shape) in enumerate(zip(lin, ((hidden, in_dim), (hidden, hidden), (bottleneck, hidden)))):
        out += [(l + ".weight", shape), (l + ".bias", (shape[0],))]
        if use_bn and i < 2:
            out += [(bnl[i] + ".weight", (hidden,)), (bnl[i] + ".bias", (hidden,))]
    return out + [(WN_G, (out_dim, 1)), (WN_V, (out_dim, bottleneck))]


def init_head_state(in_dim: int, hidden: int, bottleneck: int, out_dim: int, generator: Optional[torch.Generator] = None,
                    use_bn: bool = False) -> Dict[str, Tensor]:
    """DINOv2ProjectionHead init (dinov2_head.py:32-65): trunc-normal(0.02)/zero-bias Linear layers, weight-norm g=1,
    v = nn.Linear default init (U(-1/sqrt(fan_in), 1/sqrt(fan_in))); BatchNorm1d layers at their defaults (weight 1, bias 0,
    running_mean 0, running_var 1, num_batches_tracked 0 -- the buffers are part of the returned state, as in a state_dict)."""
    bnl = head_layer_names(use_bn)[1]
    sd: Dict[str, Tensor] = {}
    for name, shape in head_param_shapes(in_dim, hidden, bottleneck, out_dim, use_bn):
        is_bn = name.rsplit(".", 1)[0] in bnl
        if name == WN_G or (is_bn and name.endswith(".weight")):
            sd[name] = torch.ones(shape)
        elif name == WN_V:
            bound = 1 / math.sqrt(bottleneck)
            sd[name] = torch.empty(shape).uniform_(-bound, bound, generator=generator)
        elif name.endswith(".weight"):
            sd[name] = torch.nn.init.trunc_normal_(torch.empty(shape), std=0.02, generator=generator)
        else:
            sd[name] = torch.zeros(shape)
        if is_bn and name.endswith(".bias"):
            b = name.rsplit(".", 1)[0]
            sd[b + ".running_mean"], sd[b + ".running_var"] = torch.zeros(hidden), torch.ones(hidden)
            sd[b + ".num_batches_tracked"] = torch.zeros((), dtype=torch.int64)
    return sd


def vit_lr_decay_rate(name: str, lr_decay_rate: float, num_layers: int) -> float:
    """get_vit_lr_decay_rate (utils.py:155-188), un-chunked blocks."""
    layer_id = num_layers + 1
    if any(k in name for k in ("pos_embed", "patch_embed", "mask_token", "cls_token", "register_tokens")):
        layer_id = 0
    elif "blocks." in name and "residual." not in name:
        layer_id = int(name[name.find("blocks."):].split(".")[1]) + 1
    return lr_decay_rate ** (num_layers + 1 - layer_id)


def param_group_hparams(name: str, is_backbone: bool, depth: int, lr: float, args: DINOv2Args) -> Dict[str, Any]:
    """One entry of get_optimizer_with_decay's param groups (utils.py:191-250)."""
    rate = vit_lr_decay_rate(name, args.layerwise_decay, depth) if is_backbone else 1.0
    out = {"name": name, "lr": lr * rate, "weight_decay": args.weight_decay_start,
           "last_layer": "last_layer" in name, "head": "head" in name}
    if name.endswith(".bias") or "norm" in name or "gamma" in name:
        out["weight_decay"] = 0.0
    if "patch_embed" in name:
        out["lr"] = out["lr"] * args.patch_embed_lr_multiplier
    return out


def fuse_param_groups(groups: List[Dict[str, Any]]) -> List[Dict[str, Any]]:
    """get_fused_param_groups (utils.py:253-273): merge groups with identical lr / weight_decay / head / last_layer
    properties, named after (and ordered by) their first member."""
    fused: Dict[Tuple[Any, ...], Dict[str, Any]] = {}
    for g in groups:
        key = (g["lr"], g["weight_decay"], g["head"], g["last_layer"])
        if key not in fused:
            fused[key] = dict(g, names=[g["name"]])
        else:
            fused[key]["names"].append(g["name"])
    return list(fused.values())


class HeadEngine:
    """DINOv2ProjectionHead forward/backward (dinov2_head.py:66-71) on rows of bf16 features."""

    def __init__(self, params: FlatParams, prefix: str, in_dim: int, args: DINOv2Args) -> None:
        self.P, self.prefix = params, prefix
        self.in_dim, self.hid, self.bn, self.K = in_dim, args.hidden_dim, args.dino_bottleneck_dim, args.output_dim
        self.pad_wgrad_rows = True   # see backward(): weight-gradient GEMMs over a row count padded to whole K-tiles
        self.logit_dtype = torch.float32   # torch.bfloat16 under DINOv2Args.bf16_logits (set by the method object)
        dev = params.device
        # batch_norm=True (dinov2_head.py:86-92): BatchNorm1d between each hidden Linear and its GELU.  The running estimates are
        # buffers of THIS module: no gradient, not part of the EMA (update_momentum walks parameters() only), saved in the state_dict.
        self.use_bn = bool(args.batch_norm)
        self.lin, self.bnl = head_layer_names(self.use_bn)
        self.bn_eps, self.bn_momentum = 1e-5, 0.1
        from .resnet import default_bn_sync
        self.bn_sync = default_bn_sync()   # SyncBatchNorm across ranks (train_helpers.py:223): statistics of a call over every rank's rows
        self.buffers: Dict[str, Tensor] = {}
        self.batches_tracked: Dict[str, int] = {}
        for b in self.bnl:
            self.buffers[b + ".running_mean"] = torch.zeros(self.hid, dtype=torch.float32, device=dev)
            self.buffers[b + ".running_var"] = torch.ones(self.hid, dtype=torch.float32, device=dev)
            self.batches_tracked[b] = 0
        self.wn = torch.empty(self.K, self.bn, dtype=torch.bfloat16, device=dev)        # normalised prototype matrix (bf16)
        self.dwn = torch.zeros(self.K, self.bn, dtype=torch.float32, device=dev) if params.grad is not None else None

    def w(self, n: str) -> Tensor:
        return self.P.p[self.prefix + n]

    def wb(self, n: str) -> Tensor:
        return self.P.b[self.prefix + n]

    def gw(self, n: str) -> Tensor:
        return self.P.g[self.prefix + n]

    def refresh_weightnorm(self) -> None:
        ops.weightnorm_fwd(self.w(WN_V), self.w(WN_G), self.wn, self.K, self.bn)

    def load_buffers(self, state: Mapping[str, Tensor], prefix: str = "") -> None:
        """BatchNorm buffers from a head state / state_dict (keys `<prefix>mlp.N.running_mean` ...); absent keys keep their value."""
        for b in self.bnl:
            for suffix in ("running_mean", "running_var"):
                k = f"{prefix}{b}.{suffix}"
                if k in state:
                    self.buffers[f"{b}.{suffix}"].copy_(state[k].to(self.buffers[f"{b}.{suffix}"].device, torch.float32))
            k = f"{prefix}{b}.num_batches_tracked"
            if k in state:
                self.batches_tracked[b] = int(state[k])

    def buffer_state(self) -> Dict[str, Tensor]:
        out: Dict[str, Tensor] = {}
        for b in self.bnl:
            out[b + ".running_mean"] = self.buffers[b + ".running_mean"].detach().clone()
            out[b + ".running_var"] = self.buffers[b + ".running_var"].detach().clone()
            out[b + ".num_batches_tracked"] = torch.tensor(self.batches_tracked[b], dtype=torch.int64)
        return out

    def _hidden_bn(self, ws: Workspace, tag: str, li: int, xin: Tensor, k_in: int, R: int, cap: int, segs: Sequence[Tuple[int, int]],
                   training: bool) -> Dict[str, Tensor]:
        """Linear -> BatchNorm1d -> GELU on rows [0, R).  `segs` = (first row, rows) of every reference call that these rows came
        from: nn.BatchNorm1d normalises over the rows of ONE forward call, so each segment gets its own statistics, and in train()
        each moves the running estimates once, in the order given (the reference's call order)."""
        hid = self.hid
        lin, bnn = self.lin[li], self.bnl[li]
        y = ws.get(f"{tag}.y{li}", (cap, hid), torch.bfloat16)       # Linear output (BatchNorm input)
        u = ws.get(f"{tag}.u{li}", (cap, hid), torch.bfloat16)       # BatchNorm output (GELU input)
        h = ws.get(f"{tag}.h{li + 1}", (cap, hid), torch.bfloat16)
        mean = ws.get(f"{tag}.bn{li}.mean", (len(segs), hid), torch.float32)
        rstd = ws.get(f"{tag}.bn{li}.rstd", (len(segs), hid), torch.float32)
        bnws = ws.get(f"{tag}.bnws", (ops.batchnorm_ws_floats(hid),), torch.float32)
        ops.gemm(xin, self.wb(lin + ".weight"), y, M=R, N=hid, K=k_in, epilogue=ops.EPI_BF16, bias=self.w(lin + ".bias"))
        gamma, beta = self.w(bnn + ".weight"), self.w(bnn + ".bias")
        rm, rv = self.buffers[bnn + ".running_mean"], self.buffers[bnn + ".running_var"]
        for si, (r0, n) in enumerate(segs):
            if training:
                if n < 2 and self.bn_sync is None:
                    raise ValueError(f"BatchNorm1d in train() needs more than 1 row per call, got {n}")   # as torch raises
                ops.batchnorm_fwd(y[r0:r0 + n], gamma, beta, u[r0:r0 + n], mean[si], rstd[si], n, hid, bnws, running_mean=rm, running_var=rv,
                                  eps=self.bn_eps, momentum=self.bn_momentum, sync=self.bn_sync)
                self.batches_tracked[bnn] += 1
            else:
                torch.rsqrt(rv + self.bn_eps, out=rstd[si])
                ops.batchnorm_apply(y[r0:r0 + n], rm, rstd[si], gamma, beta, u[r0:r0 + n], n, hid)
        ops.gelu_fwd(u, h, R * hid)
        return dict(y=y, u=u, h=h, mean=mean, rstd=rstd)

    def forward(self, ws: Workspace, tag: str, x: Tensor, R: int, cap: int, save: bool,
                segs: Optional[Sequence[Tuple[int, int]]] = None, bn_training: bool = True) -> Dict[str, Any]:
        hid, bn, K, D = self.hid, self.bn, self.K, self.in_dim
        cap = (cap + 63) // 64 * 64   # whole 64-row tiles for the weight-gradient contractions over the rows
        l0, l1, l2 = self.lin
        bnc: List[Dict[str, Tensor]] = []
        h1p = h2p = None
        if self.use_bn:
            # with SyncBatchNorm a segment this rank has no rows for (no masked patch in its crops) still joins the collective
            segs = [(r0, n) for r0, n in (segs if segs is not None else [(0, R)]) if n > 0 or self.bn_sync is not None]
            assert sum(n for _, n in segs) == R, "BatchNorm segments must cover the rows"
            bnc.append(self._hidden_bn(ws, tag, 0, x, D, R, cap, segs, bn_training))
            h1 = bnc[0]["h"]
            bnc.append(self._hidden_bn(ws, tag, 1, h1, hid, R, cap, segs, bn_training))
            h2 = bnc[1]["h"]
        else:
            h1 = ws.get(tag + ".h1", (cap, hid), torch.bfloat16)
            h1p = ws.get(tag + ".h1p", (cap, hid), torch.bfloat16) if save else None
            ops.gemm(x, self.wb(l0 + ".weight"), h1, M=R, N=hid, K=D, epilogue=ops.EPI_BF16_GELU, bias=self.w(l0 + ".bias"), out2=h1p)
            h2 = ws.get(tag + ".h2", (cap, hid), torch.bfloat16)
            h2p = ws.get(tag + ".h2p", (cap, hid), torch.bfloat16) if save else None
            ops.gemm(h1, self.wb(l1 + ".weight"), h2, M=R, N=hid, K=hid, epilogue=ops.EPI_BF16_GELU, bias=self.w(l1 + ".bias"), out2=h2p)
        z = ws.get(tag + ".z", (cap, bn), torch.float32)
        ops.gemm(h2, self.wb(l2 + ".weight"), z, M=R, N=bn, K=hid, epilogue=ops.EPI_F32, bias=self.w(l2 + ".bias"))
        zn = ws.get(tag + ".zn", (cap, bn), torch.bfloat16)
        inv = ws.get(tag + ".inv", (cap,), torch.float32)
        ops.l2norm_fwd(z, zn, inv, R, bn, 1e-12)
        logits = ws.get(tag + ".logits", (cap, K), self.logit_dtype)
        ops.gemm(zn, self.wn, logits, M=R, N=K, K=bn, epilogue=ops.EPI_F32 if self.logit_dtype == torch.float32 else ops.EPI_BF16)
        return dict(x=x, h1=h1, h1p=h1p, h2=h2, h2p=h2p, z=z, zn=zn, inv=inv, logits=logits, R=R, cap=cap, tag=tag, bn=bnc, segs=segs)

    def backward(self, ws: Workspace, c: Dict[str, Any], dlogits: Tensor, side: Optional["torch.cuda.Stream"] = None) -> Tensor:
        """dlogits bf16 [R,K] -> returns d(x) f32 [R, in_dim]; accumulates parameter grads.

        `side`: the weight-gradient GEMMs (and `finish_weightnorm_grad`) go to this stream.  Between the cross-entropy and the start of the
        backbone's backward the step is one dependent chain of small kernels on a mostly idle chip; the weight gradients feed only the
        optimizer, so off the chain they run beside it (the 65 536 x 256 one alone is 0.3 ms).  The caller makes the optimizer -- and
        the next use of `dlogits` / the saved activations -- wait for `side`."""
        R, cap, tag = c["R"], c["cap"], c["tag"]
        hid, bn, K, D = self.hid, self.bn, self.K, self.in_dim

        slab = ws.get("wgrad.slabs", (32 * 1024 * 1024,), torch.float32)
        main = torch.cuda.current_stream() if side is not None else None

        def wgrad(dy: Tensor, xin: Tensor, out: Tensor, n_out: int, k_in: int, dbias: Optional[Tensor] = None) -> None:
            """dW[n_out, k_in] += dy[:R]^T xin[:R]; dbias[n_out] += column sums of dy[:R] (taken from the same GEMM's operand fragments).  The row count of a step (2B + local + masked rows) is data dependent and rarely a
            multiple of 64; the <= 63 rows up to the next multiple are zeroed in both operands (both: a stale pad row could hold a NaN
            bit pattern) so that the contraction runs in whole K-tiles on the 256-row slab kernel with its deterministic split-K
            reduction -- ragged, these four GEMMs fell to the 128-row kernel and fp32 atomics (1.3 ms per step at 0.3 PF/s)."""
            kpad = (R + 63) // 64 * 64
            dyp, xp = padded_rows(dy, R), padded_rows(xin, R)
            pad = self.pad_wgrad_rows and kpad != R and dyp is not None and xp is not None
            if pad:
                dy, xin = dyp, xp
            else:
                kpad = R
            tiles = ((n_out + 127) // 128) * ((k_in + 127) // 128)

            def run() -> None:
                if pad:   # (rows no data-gradient GEMM of the chain reads: zeroed on the stream of the GEMM that does)
                    dy[R:kpad].zero_()
                    xin[R:kpad].zero_()
                ops.gemm(dy, xin, out, M=n_out, N=k_in, K=kpad, trans_a=True, trans_b=True, epilogue=ops.EPI_F32_ACCUM, colsum=dbias,
                         lda=n_out, ldb=k_in, workspace=slab if kpad % 64 == 0 else None, **split_k_plan(n_out, k_in, kpad, True, _split_k(tiles, kpad)))

            if side is None:
                run()
            else:
                side.wait_event(main.record_event())
                with torch.cuda.stream(side):
                    run()

        dzn = ws.get(tag + ".dzn", (cap, bn), torch.float32)
        # [R, bn] output = only ~35 tiles but a 65 536-long contraction: split-K into slabs, accumulate into zeros
        dzn.zero_()
        # (its own slab scratch when the weight gradients run on another stream: theirs is in use there)
        dslab = slab if side is None else ws.get("head.dgrad_slabs", (16 * 1024 * 1024,), torch.float32)
        ops.gemm(dlogits, self.wn, dzn, M=R, N=bn, K=K, trans_b=True, epilogue=ops.EPI_F32_ACCUM, workspace=dslab, **split_k_plan(R, bn, K, False, 2))
        # the prototype layer's weight gradient AFTER the data gradient it shares dlogits with: on the side stream it then starts behind
        # that GEMM instead of beside it (two chip-filling launches at once made the chain's 13 us slab reduction wait 290 us for a CU)
        wgrad(dlogits, c["zn"], self.dwn, K, bn)
        dz = ws.get(tag + ".dz", (cap, bn), torch.bfloat16)
        ops.l2norm_bwd(dzn, c["z"], c["inv"], dz, R, bn)
        l0, l1, l2 = self.lin
        wgrad(dz, c["h2"], self.gw(l2 + ".weight"), bn, hid, self.gw(l2 + ".bias"))
        dh2 = ws.get(tag + ".dh2", (cap, hid), torch.bfloat16)
        dh1 = ws.get(tag + ".dh1", (cap, hid), torch.bfloat16)
        if self.use_bn:
            def bn_bwd(li: int, dh: Tensor) -> Tensor:
                """d(GELU output) -> d(Linear output): GELU', then BatchNorm backward per segment (training-mode statistics)."""
                b_, bnn = c["bn"][li], self.bnl[li]
                ops.gelu_bwd(dh, b_["u"], dh, R * hid)
                dy = ws.get(f"{tag}.dy{li}", (cap, hid), torch.bfloat16)
                bnws = ws.get(f"{tag}.bnws", (ops.batchnorm_ws_floats(hid),), torch.float32)
                for si, (r0, n) in enumerate(c["segs"]):
                    ops.batchnorm_bwd(dh[r0:r0 + n], b_["y"][r0:r0 + n], self.w(bnn + ".weight"), b_["mean"][si], b_["rstd"][si],
                                      dy[r0:r0 + n], n, hid, bnws, dgamma=self.gw(bnn + ".weight"), dbeta=self.gw(bnn + ".bias"), sync=self.bn_sync)
                return dy

            ops.gemm(dz, self.wb(l2 + ".weight"), dh2, M=R, N=hid, K=bn, trans_b=True, epilogue=ops.EPI_BF16)
            dy1 = bn_bwd(1, dh2)
            wgrad(dy1, c["h1"], self.gw(l1 + ".weight"), hid, hid, self.gw(l1 + ".bias"))
            ops.gemm(dy1, self.wb(l1 + ".weight"), dh1, M=R, N=hid, K=hid, trans_b=True, epilogue=ops.EPI_BF16)
            dh1 = bn_bwd(0, dh1)
        else:
            ops.gemm(dz, self.wb(l2 + ".weight"), dh2, M=R, N=hid, K=bn, trans_b=True, epilogue=ops.EPI_BF16_GELUGRAD, aux=c["h2p"])
            wgrad(dh2, c["h1"], self.gw(l1 + ".weight"), hid, hid, self.gw(l1 + ".bias"))
            ops.gemm(dh2, self.wb(l1 + ".weight"), dh1, M=R, N=hid, K=hid, trans_b=True, epilogue=ops.EPI_BF16_GELUGRAD, aux=c["h1p"])
        wgrad(dh1, c["x"], self.gw(l0 + ".weight"), hid, D, self.gw(l0 + ".bias"))
        dx = ws.get(tag + ".dx", (cap, D), torch.float32)
        ops.gemm(dh1, self.wb(l0 + ".weight"), dx, M=R, N=D, K=hid, trans_b=True, epilogue=ops.EPI_F32)
        return dx

    def finish_weightnorm_grad(self, side: Optional["torch.cuda.Stream"] = None) -> None:
        """(on `side` when the weight gradients ran there: it reads the prototype layer's, and feeds only the optimizer)"""
        if side is None:
            ops.weightnorm_bwd(self.dwn, self.w(WN_V), self.w(WN_G), self.gw(WN_V), self.gw(WN_G), self.K, self.bn)
            self.dwn.zero_()
            return
        with torch.cuda.stream(side):
            ops.weightnorm_bwd(self.dwn, self.w(WN_V), self.w(WN_G), self.gw(WN_V), self.gw(WN_G), self.K, self.bn)
            self.dwn.zero_()


class MockTrainerState:
    """global_step / estimated_stepping_batches as the reference reads them from Lightning's Trainer."""

    def __init__(self, total_steps: int) -> None:
        self.global_step = 0
        self.max_epochs = 1
        self.estimated_stepping_batches = total_steps


class DINOv2:
    """The method object.  Attribute / state_dict layout follows the reference."""

    supports_accumulation = True   # `accum_first` / `accum_last` / `grad_scale` are honoured by training_step_impl (subclasses: see dino.py)

    def __init__(self, vit_cfg: ViTConfig, method_args: Optional[DINOv2Args] = None, global_batch_size: int = 128,
                 total_steps: int = 125_000, device: str | torch.device = "cuda",
                 backbone_state: Optional[Dict[str, Tensor]] = None, student_head_state: Optional[Dict[str, Tensor]] = None,
                 teacher_head_state: Optional[Dict[str, Tensor]] = None, teacher_backbone_state: Optional[Dict[str, Tensor]] = None,
                 seed: int = 0, student_ibot_head_state: Optional[Dict[str, Tensor]] = None,
                 teacher_ibot_head_state: Optional[Dict[str, Tensor]] = None) -> None:
        self.method_args = method_args or DINOv2Args()
        a = self.method_args
        if a.center_method not in ("softmax", "sinkhorn_knopp"):
            raise ValueError(f"Unknown centering method: {a.center_method}")
        self.cfg = vit_cfg
        self.device = torch.device(device)
        self.global_batch_size = global_batch_size
        self.trainer = MockTrainerState(total_steps)
        g = torch.Generator().manual_seed(seed)
        D = vit_cfg.embed_dim
        bsd = backbone_state if backbone_state is not None else init_vit_state(vit_cfg, g)
        bn = a.batch_norm
        shs = student_head_state if student_head_state is not None else init_head_state(D, a.hidden_dim, a.dino_bottleneck_dim, a.output_dim, g, bn)
        ths = teacher_head_state if teacher_head_state is not None else init_head_state(D, a.hidden_dim, a.dino_bottleneck_dim, a.output_dim, g, bn)
        tbs = teacher_backbone_state if teacher_backbone_state is not None else bsd
        order_b = [n for n, _ in vit_param_shapes(vit_cfg)]
        order_h = [n for n, _ in head_param_shapes(D, a.hidden_dim, a.dino_bottleneck_dim, a.output_dim, bn)]
        s_named = [("backbone." + n, bsd[n]) for n in order_b] + [("head." + n, shs[n]) for n in order_h]
        t_named = [("backbone." + n, tbs[n]) for n in order_b] + [("head." + n, ths[n]) for n in order_h]
        if a.ibot_separate_head:
            # the reference builds the iBOT head with dino_bottleneck_dim too (dinov2.py:221-228) -- kept bug-compatible
            sis = student_ibot_head_state if student_ibot_head_state is not None else init_head_state(D, a.hidden_dim, a.dino_bottleneck_dim, a.output_dim, g, bn)
            tis = teacher_ibot_head_state if teacher_ibot_head_state is not None else init_head_state(D, a.hidden_dim, a.dino_bottleneck_dim, a.output_dim, g, bn)
            s_named += [("ihead." + n, sis[n]) for n in order_h]
            t_named += [("ihead." + n, tis[n]) for n in order_h]
        ex_s, ex_t = self._extra_params(D, g)      # further trained modules of a subclass (DINOv31: the PaKA heads), same layout on both sides
        s_named += ex_s
        t_named += ex_t
        self.student = FlatParams(s_named, self.device, True)
        self.teacher = FlatParams(t_named, self.device, False)
        if self.world > 1:
            # what DDP does when it wraps the module: every replica starts from rank 0's parameters (ranks built from different
            # seeds / states would otherwise diverge silently)
            for fp in (self.student, self.teacher):
                dist.broadcast(fp.data, src=0)
                fp.bf16.copy_(fp.data)
        self.s_vit = ViTEngine(vit_cfg, self.student, "backbone.")
        self.t_vit = ViTEngine(vit_cfg, self.teacher, "backbone.")
        self.s_head = HeadEngine(self.student, "head.", D, a)
        self.t_head = HeadEngine(self.teacher, "head.", D, a)
        self.s_ihead = HeadEngine(self.student, "ihead.", D, a) if a.ibot_separate_head else self.s_head
        self.t_ihead = HeadEngine(self.teacher, "ihead.", D, a) if a.ibot_separate_head else self.t_head
        for h in {id(x): x for x in (self.s_head, self.t_head, self.s_ihead, self.t_ihead)}.values():
            h.refresh_weightnorm()
        self.s_head.load_buffers(shs)
        self.t_head.load_buffers(ths)
        if a.ibot_separate_head:
            self.s_ihead.load_buffers(sis)
            self.t_ihead.load_buffers(tis)
        # freeze_eval_module(self.teacher_head) (dinov2.py:63-67,241) leaves the teacher heads in eval(): their BatchNorm layers apply
        # the running estimates, which nothing ever moves (no forward in train(), not in the EMA).  A Trainer that calls
        # module.train() at fit start (Lightning < 2.2) flips them to batch statistics: set this attribute to get that behaviour.
        self.teacher_head_training = False
        K = a.output_dim
        self.dino_center = torch.zeros(1, K, device=self.device)
        self.ibot_center = torch.zeros(1, 1, K, device=self.device)
        self._pending: Dict[str, Tuple[Tensor, float]] = {}
        self.ws = Workspace(self.device)
        # optimizer state
        self.exp_avg = torch.zeros_like(self.student.data)
        self.exp_avg_sq = torch.zeros_like(self.student.data)
        self.opt_step = 0
        lr_scale = global_batch_size / a.reference_batch_size
        if a.lr_scale_method == "sqrt":
            lr_scale = math.sqrt(lr_scale)
        self.base_lr = a.lr * lr_scale
        self.param_groups: List[Dict[str, Any]] = []
        for n in self.student.names:
            is_bb = n.startswith("backbone.")
            if is_bb:
                ref_name = n[len("backbone."):]
            elif n.startswith(("head.", "ihead.")):
                ref_name = "ibot_head." + n[len("ihead."):] if n.startswith("ihead.") else "dino_head." + n[len("head."):]
            else:   # a subclass's extra module: named_parameters() of the bare module (get_optimizer_with_decay walks trainable modules)
                ref_name = n.split(".", 1)[1]
            self.param_groups.append(param_group_hparams(ref_name, is_bb, vit_cfg.depth, self.base_lr, a))
        dev = self.device
        self.seg_lr = torch.tensor([g_["lr"] for g_ in self.param_groups], dtype=torch.float32, device=dev)
        self.seg_wd_on = torch.tensor([1 if g_["weight_decay"] != 0.0 else 0 for g_ in self.param_groups], dtype=torch.uint8, device=dev)
        # freeze masks of on_before_optimizer_step (dinov2.py:619-635): bit 0 = "last_layer" groups, bit 1 = groups without "head" in
        # their name (the backbone)
        self.seg_frozen = torch.tensor([(1 if g_["last_layer"] else 0) | (0 if g_["head"] else 2) for g_ in self.param_groups],
                                       dtype=torch.uint8, device=dev)
        self.warmup_steps = min(total_steps - 1, a.warmup_steps)
        self._sumsq = torch.zeros(1, device=dev)
        self._loss_slots = torch.zeros(5, device=dev)   # weighted dino_global, dino_local, ibot, koleo; [4] = unweighted KoLeo value at weight 0
        self._static_idx: Dict[Tuple[int, ...], Dict[str, Tensor]] = {}
        self._dxn_rows: Dict[str, Tuple[int, Tuple[int, ...], Tensor]] = {}   # per pass: (buffer, shape, rows the last step wrote) of the upstream-gradient buffer
        self.last_grad_norm: Optional[Tensor] = None
        self.overlap_streams = True
        # order-fixed reductions (csrc/reduce.hip): LayerNorm / bias / LayerScale / mask-token gradients without fp32 atomics, so that a
        # step is bitwise reproducible; LT_DETERMINISTIC=0 goes back to atomics
        self.deterministic = os.environ.get("LT_DETERMINISTIC", "1") != "0"
        self.two_bwd_chains = os.environ.get("LT_BWD_TWO_CHAINS", "1") != "0"
        # forward concurrency: the local-crop pass on the side stream beside the global-crop pass, the teacher on a stream of its own
        # (0: that pass runs on the main stream -- tools/ab_schedule.py measures the fewer-streams corners)
        self.fwd_local_stream = int(os.environ.get("LT_FWD_LOCAL_STREAM", "1") != "0")
        self.fwd_teacher_stream = int(os.environ.get("LT_FWD_TEACHER_STREAM", "1") != "0")
        # HIP-graph replay of the static blocks of the backward pass (`_backward_backbone`); the forward's switch lives on the ViT engines
        self.graph_backward = int(os.environ.get("LT_GRAPH_BWD", "0") != "0")
        # launch-plan replay of the same static blocks (round 6, ops.LaunchPlan; the default): the calls across the C ABI and the event edges of
        # blocks depth-2 .. 0 of both chains and the weight-gradient stream, logged on the third step of a geometry and replayed by a bare
        # loop afterwards -- the launches, their order and their streams are the eager step's
        self.plan_backward = int(os.environ.get("LT_PLAN_BWD", "1") != "0") and not self.graph_backward
        self._bwd_graph: Dict[str, Any] = {}
        self._graph_chain_stream: Optional["torch.cuda.Stream"] = None
        # one weight-gradient GEMM per layer for the global- and the local-crop pass (vit.JointWgrad): half the split-K slab traffic
        self.joint_wgrad = int(os.environ.get("LT_JOINT_WGRAD", "1") != "0")
        self.head_side = int(os.environ.get("LT_HEAD_SIDE", "1") != "0")   # KoLeo and the heads' weight gradients beside the head chain
        self._joint: Optional[JointWgrad] = None
        self._joint_active: Optional[JointWgrad] = None
        # the last block's MLP branch, forward and backward, only at the token rows the losses read (cls + masked patches): vit.forward
        self.sparse_last_mlp = os.environ.get("LT_SPARSE_LAST_MLP", "1") != "0"
        # softmax centering without the [rows, K] probability matrix (training_step_impl); LT_FUSED_CENTERING=0: softmax, column sums and
        # cross-entropy as three passes
        self.fused_centering = os.environ.get("LT_FUSED_CENTERING", "1") != "0"
        env_bl = os.environ.get("LT_BF16_LOGITS")
        self.bf16_logits = (a.bf16_logits if env_bl is None else env_bl != "0") and a.center_method == "softmax" and self.fused_centering
        if self.bf16_logits:
            for h_ in (self.s_head, self.t_head, self.s_ihead, self.t_ihead):
                h_.logit_dtype = torch.bfloat16
        self.sinkhorn_joint = os.environ.get("LT_SINKHORN_JOINT", "1") != "0"   # both heads' Sinkhorn iterations share one all-reduce each
        # gradient accumulation (the reference hands `gradient_accumulation_steps` to Lightning as accumulate_grad_batches,
        # LT/_commands/train_helpers.py:224-236): a caller that accumulates k micro-batches per optimizer step sets, before each
        # `training_step_impl`, accum_first (first micro-batch of the window: the flat gradient buffer is zeroed), grad_scale = 1 / k
        # (Lightning divides the loss by k before backward) and accum_last (False: no gradient collective is started under this
        # backward and the LayerScale gradients, which are formed from the ACCUMULATED weight gradients, wait for `optimizer_step`)
        self.accum_first, self.accum_last, self.grad_scale = True, True, 1.0
        self._ls_finished = True
        # reference _activation_checkpointing.py / DINOv2ViTModelWrapper: keep only block inputs of the student, recompute each
        # block in backward (+1 student forward, ~9x less activation memory); off by default -- 288 GB rarely needs it
        self.activation_checkpointing = False
        # host RNG of the stochastic-depth draws: per-rank stream (the reference draws from each rank's device RNG)
        rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
        self._drop_gen = torch.Generator().manual_seed(seed + 7919 + 104729 * rank)
        self._grad_sync: Optional[GradSync] = None
        # set to a list to collect (start, end) HIP events around the point where the optimizer has to wait for the gradient
        # all-reduces: their elapsed time is the EXPOSED (not hidden under backward) communication time of a step
        self.comm_events: Optional[List[Tuple[Any, Any]]] = None
        # data parallel: all-reduce the head gradients and each transformer block's gradients as soon as they are final,
        # underneath the rest of backward (what DDP's bucket hooks do in the reference); LT_GRAD_OVERLAP=0 reduces after it
        self.overlap_grad_reduce = os.environ.get("LT_GRAD_OVERLAP", "1") != "0"
        # iBOT masks of the coming steps sampled in a background process (same `random` stream as the in-line call); off by
        # default so that `random.seed()` between steps keeps its in-line meaning -- a training loop switches it on once
        self.prefetch_masks = False
        self._mask_producer: Optional[MaskProducer] = None
        self._head_span = self.student.span(("head.", "ihead."))
        self._block_spans = [self.student.span((f"backbone.blocks.{i}.",)) for i in range(vit_cfg.depth)]
        use_streams = self.device.type == "cuda"
        self.side_stream = torch.cuda.Stream(device=self.device) if use_streams else None     # weight-gradient GEMMs
        self.teacher_stream = torch.cuda.Stream(device=self.device) if use_streams else None  # teacher forward
        self.local_bwd_stream = torch.cuda.Stream(device=self.device) if use_streams else None  # local-crop dgrad chain
        self.reduce_stream = torch.cuda.Stream(device=self.device) if use_streams else None    # orders the early all-reduces

    def _extra_params(self, D: int, g: torch.Generator) -> Tuple[List[Tuple[str, Tensor]], List[Tuple[str, Tensor]]]:
        """(student, teacher) lists of further named parameters behind the heads (none for DINOv2)."""
        return [], []

    # ------------------------------------------------------------------ reference-compatible views
    def close(self) -> None:
        """Stop background helpers (the mask-sampling process of `prefetch_masks`)."""
        if self._mask_producer is not None:
            self._mask_producer.close()
            self._mask_producer = None

    def state_dict(self) -> Dict[str, Tensor]:
        """`method.state_dict()` with the reference's keys (what Lightning stores as checkpoint["state_dict"])."""
        buffers = {("student", "head."): self.s_head.buffer_state(), ("teacher", "head."): self.t_head.buffer_state(),
                   ("student", "ihead."): self.s_ihead.buffer_state(), ("teacher", "ihead."): self.t_ihead.buffer_state()}
        return checkpoint.method_state_dict(self.student, self.teacher, {"dino_loss.center": self.dino_center, "ibot_loss.center": self.ibot_center},
                                            self.method_args.ibot_separate_head, self.cfg.depth, self.cfg.block_chunks, buffers)

    def reset_transient_buffers(self) -> None:
        """Forget everything earlier steps left in the reused activation buffers: the allocation-time zero fills are renewed and the
        incremental clearing of the upstream-gradient buffers starts over with a full fill.  Called by `load_state_dict` (so by every
        resume / in-process recovery); call it directly after a step whose loss or gradient norm came out non-finite if the weights are
        restored by other means."""
        self.ws.rezero()
        self._dxn_rows = {}

    def load_state_dict(self, sd: Mapping[str, Tensor], strict: bool = True) -> None:
        """Load a `method.state_dict()` written by the reference (or by `state_dict()`): fp32 master weights, their bf16 shadows
        and every derived cache (weight-normed prototype matrices, padded patch-embedding matrices), loss centers included.
        Pending (not yet applied) center updates of the running step are dropped, as in a freshly constructed reference module."""
        extra = checkpoint.load_method_state_dict(sd, self.student, self.teacher, self.method_args.ibot_separate_head, strict)
        self.reset_transient_buffers()
        if "dino_loss.center" in extra:
            self.dino_center.copy_(extra["dino_loss.center"].to(self.device, torch.float32).view_as(self.dino_center))
        if "ibot_loss.center" in extra:
            self.ibot_center.copy_(extra["ibot_loss.center"].to(self.device, torch.float32).view_as(self.ibot_center))
        for role, head, ihead in (("student", self.s_head, self.s_ihead), ("teacher", self.t_head, self.t_ihead)):
            head.load_buffers(extra, f"{role}_head.dino_head.")
            if ihead is not head:
                ihead.load_buffers(extra, f"{role}_head.ibot_head.")
        self._pending.clear()
        self._refresh_derived()

    def _refresh_derived(self) -> None:
        for h in {id(x): x for x in (self.s_head, self.t_head, self.s_ihead, self.t_ihead)}.values():
            h.refresh_weightnorm()
        self.s_vit.refresh_padded_weights()
        self.t_vit.refresh_padded_weights()

    def _group_entries(self) -> List[Dict[str, Any]]:
        """Per-tensor optimizer entries in the reference's `named_parameters()` order, with the values currently in effect."""
        a, k, total = self.method_args, self.trainer.global_step, self.trainer.estimated_stepping_batches
        lr_factor = warmup_cosine_lr_factor(k, self.warmup_steps, total, a.min_lr / self.base_lr)
        wd = cosine_schedule(min(max(k - 1, 0), total), total, a.weight_decay_start, a.weight_decay_end) if k > 0 else a.weight_decay_start
        out = []
        for n, g_ in zip(self.student.names, self.param_groups):
            out.append(dict(g_, flat=n, lr_now=g_["lr"] * lr_factor, wd_now=wd if g_["weight_decay"] != 0.0 else 0.0))
        return out

    def optimizer_state_dict(self) -> Dict[str, Any]:
        """`torch.optim.AdamW.state_dict()` of the reference's optimizer (fused groups of utils.py:191-273) from the flat moments."""
        a = self.method_args
        hyper = dict(betas=tuple(a.betas), eps=a.eps, amsgrad=False, maximize=False, foreach=True, capturable=False, differentiable=False,
                     fused=None, decoupled_weight_decay=True)
        return checkpoint.optimizer_state_dict(self.student, self.exp_avg, self.exp_avg_sq, self.opt_step, self._group_entries(), hyper)

    def load_optimizer_state_dict(self, osd: Mapping[str, Any]) -> None:
        self.opt_step = checkpoint.load_optimizer_state_dict(osd, self.student, self.exp_avg, self.exp_avg_sq, self._group_entries())

    def lr_scheduler_state(self) -> Dict[str, Any]:
        """`CosineWarmupScheduler.state_dict()` (a LambdaLR: every attribute but the optimizer; the lambda is stored as None) at the
        current step, for the optimizer's parameter groups as `optimizer_state_dict()` lists them."""
        k = self.trainer.global_step
        groups = self.optimizer_state_dict()["param_groups"]
        return {"warmup_epochs": self.warmup_steps, "max_epochs": int(self.trainer.estimated_stepping_batches), "start_value": 1.0,
                "end_value": self._lr_end_value(), "period": None, "base_lrs": [g_.get("initial_lr", g_["lr"]) for g_ in groups], "last_epoch": k,
                "verbose": False, "_step_count": k + 1, "_get_lr_called_within_step": False, "_last_lr": [g_["lr"] for g_ in groups],
                "lr_lambdas": [None]}

    def _lr_end_value(self) -> float:
        return self.method_args.min_lr / self.base_lr       # configure_optimizers (dinov2.py:561-572)

    def checkpoint_dict(self) -> Dict[str, Any]:
        """The parts of a Lightning checkpoint this step owns (LT/_checkpoint.py:101-123; Lightning's `dump_checkpoint`):
        module state, optimizer state, scheduler state, global step; `epoch` from `trainer.steps_per_epoch` when the caller set it
        (Lightning writes its own when it drives the loop: integration.py)."""
        spe = getattr(self.trainer, "steps_per_epoch", None)
        return {"state_dict": self.state_dict(), "optimizer_states": [self.optimizer_state_dict()], "lr_schedulers": [self.lr_scheduler_state()],
                "global_step": self.trainer.global_step, "epoch": (self.trainer.global_step // spe) if spe else 0}

    def load_checkpoint_dict(self, ckpt: Mapping[str, Any], strict: bool = True) -> None:
        """Resume: inverse of `checkpoint_dict()`; also accepts a checkpoint written around the reference's own module.  The optimizer's
        step count and the checkpoint's global_step must agree (one optimizer step per batch in this method)."""
        self.load_state_dict(ckpt["state_dict"], strict=strict)
        if ckpt.get("optimizer_states"):
            self.load_optimizer_state_dict(ckpt["optimizer_states"][0])
        gs = int(ckpt.get("global_step", self.opt_step))
        if ckpt.get("optimizer_states") and self.opt_step not in (0, gs) and self._opt_counts_steps():
            raise ValueError(f"checkpoint global_step {gs} does not match its optimizer step count {self.opt_step}")
        self.trainer.global_step = gs

    def _opt_counts_steps(self) -> bool:
        return True

    def export_backbone_state_dict(self) -> Dict[str, Tensor]:
        """What the reference exports (EMA teacher backbone, dinov2_vit_package.py:146-162)."""
        return {checkpoint.vit_key_from_flat(n[9:], self.cfg.depth, self.cfg.block_chunks): self.teacher.p[n].detach().clone()
                for n in self.teacher.names if n.startswith("backbone.")}

    @property
    def world(self) -> int:
        return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1

    # ------------------------------------------------------------------ helpers
    def _apply_center_updates(self) -> None:
        m = self.method_args.center_momentum
        for key, center in (("dino", self.dino_center), ("ibot", self.ibot_center)):
            if key in self._pending:
                colsum, scale, handle = self._pending.pop(key)
                if handle is not None:
                    handle.wait()
                ops.center_ema(center.view(-1), colsum, scale / self.world, m, center.numel())

    def _indices(self, B: int, n_p_g: int, n_local: int, n_p_l: int) -> Dict[str, Tensor]:
        key = (B, n_p_g, n_local, n_p_l)
        if key not in self._static_idx:
            Ng, Nl = n_p_g + 1, n_p_l + 1
            r = torch.arange(2 * B)
            d = dict(
                t_cls=(((r + B) % (2 * B)) * Ng).to(torch.int64),   # teacher cls rows, halves swapped (dinov2.py:414-420)
                s_cls=(r * Ng).to(torch.int64),
                l_cls=(torch.arange(n_local * B) * Nl).to(torch.int64),
            )
            self._static_idx[key] = {k: v.to(self.device) for k, v in d.items()}
        return self._static_idx[key]

    def _reduce_begin(self) -> None:
        """Open the reduction ledger for this step's backward (before the projection-head backward); `_backward_backbone` closes it."""
        if getattr(self, "_grad_zeroed", None) is not None:    # the gradient buffer was zeroed on the side stream: order this stream's writers after it
            torch.cuda.current_stream().wait_event(self._grad_zeroed)
            self._grad_zeroed = None
        if not self.deterministic or self.device.type != "cuda":
            return
        cfg, a = self.cfg, self.method_args
        D = cfg.embed_dim
        H1 = int(D * cfg.mlp_ratio) * (2 if cfg.swiglu else 1)
        per_pass = 2 * 256 * 3 * D + 128 * (3 * D + D + H1 + D) + 4 * 64 * D     # LayerNorm x 2, four bias sums, LayerScale x 2 of one block
        floats = 2 * (cfg.depth * per_pass + 256 * 3 * D + 2 * 1024 * D) + 4 * 128 * (2 * a.hidden_dim + 2 * D + 4096)
        # the bias column sums that ride the weight-gradient GEMMs reserve (k-slices x column tiles x 4) partial rows of n_out floats each;
        # the split-K plan keeps row tiles x k-slices within 512 workgroup slots, so a GEMM needs at most 2048 / (n_out / 256) rows, i.e.
        # 2048 * 256 floats whatever its shape: four such GEMMs per block and pass (two passes, or their joint form), plus the heads'
        # (projection MLP x 3, PaKA head x 3 in DINOv31, both iBOT / DINO heads).  HBM is not the scarce resource here (288 GB).
        floats += (2 * 4 * cfg.depth + 12) * 2048 * 256
        self._reduce_floats = int(floats * 1.25) // 4 * 4
        ops.reduce_begin(self._reduce_scratch("reduce.scratch"))

    def _reduce_scratch(self, name: str) -> Tensor:
        return self.ws.get(name, (self._reduce_floats,), torch.float32)

    def _backward_backbone(self, sg: Dict[str, Any], dxn_g: Tensor, sl: Optional[Dict[str, Any]], dxn_l: Optional[Tensor]) -> None:
        """Backward of the student ViT from the gradients at its final-norm output: global crops (`sg`) and local crops (`sl`), with the
        early gradient all-reduces of data-parallel runs.  The projection-head gradients are final when this is called."""
        ws = self.ws
        main = torch.cuda.current_stream()
        side = self.side_stream if self.overlap_streams else None
        sync = self._gradient_sync() if self.overlap_grad_reduce and self.reduce_stream is not None and self.accum_last else None
        done_blocks: List[int] = []

        def reduce_block(i: int, after: Tuple[Any, ...]) -> None:
            """Block i is through every backward pass: finish its LayerScale gradients and start its all-reduce, ordered
            after the streams that wrote its gradients, on a stream of its own (nothing of backward waits for it)."""
            if sync is None:
                return
            rs = self.reduce_stream
            for st in after:
                if st is not None:
                    rs.wait_event(st.record_event())
            with torch.cuda.stream(rs):
                if det:
                    ops.reduce_flush()    # every producer enqueued so far is on a stream `rs` has just waited for
                self.s_vit.finish_layerscale_grads(blocks=[i], last_call=False)
                sync.start(*self._block_spans[i])
            done_blocks.append(i)

        det = self.deterministic and self.device.type == "cuda"
        if sync is not None:               # the prototype heads are final: their all-reduce runs under the whole ViT backward,
            rs0 = self.reduce_stream       # ordered after the streams that wrote their gradients (the side stream holds the weight gradients)
            rs0.wait_event(main.record_event())
            if side is not None:
                rs0.wait_event(side.record_event())
            with torch.cuda.stream(rs0):
                if det:
                    ops.reduce_flush()     # the head's bias sums
                sync.start(*self._head_span)
        if sl is not None and side is not None and self.local_bwd_stream is not None and self.two_bwd_chains:
            # two independent dgrad chains (local / global crops) on two streams, launches interleaved block by block; the
            # weight-gradient GEMMs of both go to `side` in that order (ordered read-modify-writes of the shared gradient)
            lstream2 = self.local_bwd_stream
            lstream2.wait_event(main.record_event())
            jw = self._joint_active
            depth = self.cfg.depth
            # HIP-graph replay of the static blocks depth-2 .. 0 (all but the first one processed, which runs on the rows the losses read):
            # both dgrad chains, the weight-gradient stream and their event choreography become ONE graph launch.  Eligible when nothing
            # in those blocks depends on the step (no stochastic-depth draws, no rotary draws, no early all-reduces, no recomputation).
            plain = lambda c: all(b[k]["mode"] == "plain" and b[k]["rowscale"] is None for b in c["blocks"][:-1] for k in ("attn", "mlp"))   # noqa: E731
            gkey = None
            use_plan = bool(self.plan_backward) and not self.graph_backward and ops.plan_replay_enabled
            if ((self.graph_backward or use_plan) and sync is None and depth > 1 and self.device.type == "cuda" and not self.activation_checkpointing
                    and sl.get("rope") is None and sl.get("block_in") is None and sg.get("block_in") is None and plain(sl) and plain(sg)):
                gkey = (sg["T"], sl["T"], jw is not None, det, dxn_g.data_ptr(), dxn_l.data_ptr(), ws.generation, use_plan,
                        main.cuda_stream, lstream2.cuda_stream, side.cuda_stream)
            G = self._bwd_graph
            if gkey is None or G.get("key") != gkey:
                G.clear()
                G.update(key=gkey, calls=0, graph=None, state=None)
            G["calls"] += 1
            replaying = gkey is not None and G["graph"] is not None
            stop = 1 if replaying else None
            chains = [(lstream2, self.s_vit.backward_iter(ws, sl, dxn_l, side=side, joint=jw, stop_after=stop)),
                      (main, self.s_vit.backward_iter(ws, sg, dxn_g, side=side, joint=jw, stop_after=stop))]
            live = [True, True]

            def one_iteration() -> None:
                for ci, (st, gen) in enumerate(chains):
                    if live[ci]:
                        with torch.cuda.stream(st):
                            live[ci] = next(gen) != "tail"
                if jw is not None:
                    jw.flush()       # a layer only one of the passes ran on all rows

            def boundary() -> None:
                """Between the eagerly launched first block and the (captured / replayed / eagerly launched) static blocks: the main stream
                joins the other two, the ledger's first region is summed, and a region of its own opens for the static blocks -- their
                partial rows live at addresses the graph holds, which no eagerly launched kernel of any later step may be handed."""
                main.wait_stream(lstream2)
                main.wait_stream(side)
                # (An event-only boundary -- the chains' buffer-reuse events replaced by one per-step event of the weight-gradient stream, the
                # first region summed on an idle stream, no stream waiting for another -- was built and measured: 82.8-85.2 ms against
                # 79.1-79.4 with this join and 78.9-79.3 eager, profiles/r06j_plan_boundary_ab.log; removed.)
                # ... and the local-crop chain joins them too: its map of "the weight-gradient stream still reads this buffer" events is
                # dropped below, so the join has to stand in for them (without it, block depth-2's attention backward on that chain could
                # overwrite the qkv gradient the joint weight-gradient GEMM of block depth-1 was still reading: seen once in a full-suite run)
                lstream2.wait_event(main.record_event())
                for c in (sl, sg):
                    c["_bwd_consumed"].clear()     # every event in there is behind the join
                if det:
                    ops.reduce_flush()
                    ops.reduce_begin(self._reduce_scratch("reduce.scratch_graph"), 64)

            def after_static(stream_now: "torch.cuda.Stream") -> None:
                """End of the static blocks, on a stream that is behind all three: their region's ordered sums."""
                if det:
                    ops.reduce_flush()

            blk = depth
            if gkey is None:
                while any(live):
                    one_iteration()
                    blk -= 1
                    if blk >= 0:
                        reduce_block(blk, (lstream2, main, side))
            else:
                one_iteration()                       # block depth-1: eager in every step
                boundary()
                if replaying:
                    G["graph"].replay()
                    G["replays"] = G.get("replays", 0) + 1
                elif G["calls"] < 2:                  # first step of this geometry: the same structure, launched eagerly (allocates)
                    for _ in range(depth - 1):
                        one_iteration()
                    main.wait_stream(lstream2)
                    main.wait_stream(side)
                    after_static(main)
                elif use_plan:                        # second step: the same launches on the same streams, logged as they are made
                    with ops.record_plan() as lp:
                        for _ in range(depth - 1):
                            one_iteration()
                        for c in (sl, sg):
                            c["_bwd_consumed"].clear()
                        main.wait_stream(lstream2)
                        main.wait_stream(side)
                        after_static(main)
                    if ws.generation == gkey[6]:
                        G["graph"] = lp
                else:                                 # second step: capture, then run it
                    g = torch.cuda.CUDAGraph()
                    # The capture's ORIGIN is the weight-gradient stream; both dgrad chains are streams forked off it (the global-crop
                    # chain on a stream of its own: the caller's stream may be the legacy default stream, which cannot take part).
                    # ROCm 7.0's capture dies in hipStreamEndCapture when a forked stream waits for an event of another forked stream
                    # that depends on it (chain -> weight-gradient stream -> same chain); edges to and from the origin are fine
                    # (tools/graph_min_probe.py v6 / v7 / v10, profiles/r05_graph_capture_probe.log).
                    if self._graph_chain_stream is None:
                        self._graph_chain_stream = torch.cuda.Stream(device=self.device)
                    gstream = self._graph_chain_stream
                    with torch.cuda.graph(g, stream=side):
                        chains[1] = (gstream, chains[1][1])
                        for st in (lstream2, gstream):
                            st.wait_stream(side)      # fork
                        for _ in range(depth - 1):
                            one_iteration()
                        for c in (sl, sg):
                            c["_bwd_consumed"].clear()
                        for st in (lstream2, gstream):
                            side.wait_stream(st)      # join
                        after_static(side)
                    chains[1] = (main, chains[1][1])
                    torch.cuda.set_stream(main)
                    g.replay()
                    G["graph"] = g
                if det:                               # third region: the tails (and whatever follows until reduce_end)
                    ops.reduce_begin(self._reduce_scratch("reduce.scratch"), 128)
                if replaying:
                    chains = [(lstream2, self.s_vit.backward_iter(ws, sl, dxn_l, side=side, joint=jw, resume=G["state"][0])),
                              (main, self.s_vit.backward_iter(ws, sg, dxn_g, side=side, joint=jw, resume=G["state"][1]))]
                    live = [True, True]
                one_iteration()                       # both generators arrive at "tail" (nothing is launched)
                assert not any(live)
                if G["graph"] is not None and G["state"] is None:
                    G["state"] = (sl["_bwd_state"], sg["_bwd_state"])
            main.wait_stream(lstream2)
            for _, gen in chains:   # tails: plain accumulations into cls/pos/patch-embedding gradients, one after the other
                for _ in gen:
                    pass
        else:
            if sl is not None:
                self.s_vit.backward(ws, sl, dxn_l, side=side)
            blk = self.cfg.depth
            for ev in self.s_vit.backward_iter(ws, sg, dxn_g, side=side):
                if ev == "block":
                    blk -= 1
                    reduce_block(blk, (main, side))
        if side is not None:
            main.wait_stream(side)
        if sync is not None:
            main.wait_stream(self.reduce_stream)
        if det:
            ops.reduce_end()               # one ordered sum for everything still recorded; back to immediate reductions
            n_ovf = ops.reduce_overflows()
            if n_ovf != getattr(self, "_ledger_overflows", 0):   # scratch too small for this shape: those sums fell back to atomics
                import warnings
                warnings.warn(f"reduction ledger scratch exhausted ({n_ovf} fallbacks to atomic sums so far): the step is not bitwise reproducible")
                self._ledger_overflows = n_ovf
        # LayerScale gradients come from the accumulated weight gradients (dgamma = (W . dW + b db) / gamma, added once): inside an
        # accumulation window they wait for the last micro-batch -- or for `optimizer_step` when the caller did not know it was the last
        self._ls_finished = bool(self.accum_last)
        if self.accum_last:
            self.s_vit.finish_layerscale_grads(blocks=[i for i in range(self.cfg.depth) if i not in done_blocks])

    # ------------------------------------------------------------------ the step
    def training_step_impl(self, batch: Mapping[str, Any], batch_idx: int, masks: Optional[Dict[str, Tensor]] = None) -> TrainingStepResult:
        a, cfg, ws, dev = self.method_args, self.cfg, self.ws, self.device
        step = self.trainer.global_step
        teacher_temp = linear_warmup_schedule(step, a.teacher_temp_warmup_steps, a.teacher_temp_start, a.teacher_temp_end)
        views: List[Tensor] = batch["views"]
        n_global = 2
        n_local = len(views) - n_global
        terms = (n_global - 1) * n_global + max(n_local * n_global, 1)
        views = [v.to(dev, torch.float32, non_blocking=True) for v in views]   # pinned host views: async H2D per view, cat on the GPU
        gv = torch.cat(views[:n_global])
        n_crops = gv.shape[0]
        B = n_crops // n_global
        p = cfg.patch_size
        gh, gw = -(-gv.shape[2] // p), -(-gv.shape[3] // p)  # ceil: PatchEmbed pad-resizes to the next multiple of p
        n_p = gh * gw
        n_reg = cfg.num_register_tokens
        Ng = n_p + 1 + n_reg   # tokens per global crop: [cls | registers | patches]
        D, K = cfg.embed_dim, a.output_dim

        if masks is None and isinstance(batch, Mapping) and batch.get("masks") is not None:
            masks = batch["masks"]   # injected iBOT masks (parity tests drive the Method hook, whose signature has no masks argument)
        if masks is None and self.prefetch_masks:
            key = (a.mask_ratio_min, a.mask_ratio_max, int(n_crops * a.mask_probability), n_crops, (gh, gw))
            if self._mask_producer is None or self._mask_producer.key != key:   # first step, or the batch geometry changed
                if self._mask_producer is not None:
                    self._mask_producer.close()
                self._mask_producer = MaskProducer(*key[:4], grid=(gh, gw))
            masks = self._mask_producer.get()
        elif masks is None:
            gen = MaskingGenerator(input_size=(gh, gw), max_num_patches=int(0.5 * gh * gw))
            masks = create_collated_masks(a.mask_ratio_min, a.mask_ratio_max, int(n_crops * a.mask_probability), n_crops, gen)
        cm = masks["collated_masks"]
        # (bool -> uint8 as a reinterpretation: a converting copy of > 32 768 elements opens an OpenMP region on the launch thread)
        mask_u8 = ops.h2d(cm.view(torch.uint8) if cm.dtype == torch.bool else cm.to(torch.uint8), dev)
        midx = masks["mask_indices_list"].to(torch.int64)
        M = int(midx.shape[0])
        mw = masks["masks_weight"].to(torch.float32)
        patch_rows = ops.h2d((midx // n_p) * Ng + 1 + n_reg + midx % n_p, dev)
        # a masked crop holds at most int(n_p * ratio_max) patches (MaskingGenerator.__call__ caps the total, utils.py:120-152);
        # the 0.5 of `max_num_patches` only bounds one block.  Buffers are sized for the worst case of this configuration.
        cap_M = int(n_crops * a.mask_probability) * max(int(n_p * max(a.mask_ratio_max, 0.5)), 1)
        if M > cap_M:
            raise ValueError(f"{M} masked patches exceed the capacity {cap_M} implied by mask_ratio_max={a.mask_ratio_max}")

        lv = torch.cat(views[n_global:]).to(dev, torch.float32) if n_local > 0 else None
        n_p_l = (-(-lv.shape[2] // p)) * (-(-lv.shape[3] // p)) if lv is not None else 0
        Nl = n_p_l + 1 + n_reg
        ix = self._indices(B, n_p + n_reg, n_local, n_p_l + n_reg)
        gs = float(self.grad_scale)
        if self.accum_first:
            if self._grad_sync is not None:
                self._grad_sync.reset()   # a step whose optimizer_step was skipped must not leak its ranges into this one
            # (on the side stream, which also carries every weight-gradient accumulation: the 344 MB fill runs beside the view concatenation
            # instead of ahead of the whole step; the first gradient writer of the other streams waits for it in `_reduce_begin`)
            zs = self.side_stream if (self.side_stream is not None and self.overlap_streams and os.environ.get("LT_GRAD_ZERO_SIDE", "1") != "0") else None
            self._grad_zeroed = None
            if zs is None:
                self.student.grad.zero_()
            else:
                zs.wait_event(torch.cuda.current_stream().record_event())
                with torch.cuda.stream(zs):
                    self.student.grad.zero_()
                    self._grad_zeroed = zs.record_event()
        self._loss_slots.zero_()

        # the losses read the final tokens at the cls rows and the masked patch rows only: the last block's MLP branch runs there alone
        # (index tensors built on the main stream, before the teacher stream forks off it)
        rows_g = (torch.cat([ix["s_cls"][:2 * B], patch_rows[:M]]), 2 * B + M) if self.sparse_last_mlp else None
        rows_l = (ix["l_cls"], n_local * B) if (self.sparse_last_mlp and n_local > 0) else None
        # ---------------- teacher (no grad) : dinov2.py:399-472 -- on its own stream, concurrent with the student forward
        main = torch.cuda.current_stream()
        # the teacher and the student's global pass unfold the same images: one patch matrix for both (the iBOT masks act on the tokens)
        shared_cols = ops.im2col(gv.contiguous(), p, self.s_vit.kpad) if (gv.shape[2] % p == 0 and gv.shape[3] % p == 0 and dev.type == "cuda"
                                                                           and os.environ.get("LT_SHARED_COLS", "1") != "0") else None
        tstream = self.teacher_stream if (self.teacher_stream is not None and self.overlap_streams and self.fwd_teacher_stream) else main
        tstream.wait_event(main.record_event())
        torch.cuda.set_stream(tstream)
        if a.center_method == "softmax":
            self._apply_center_updates()
        tctx = self.t_vit.forward(ws, "t", gv, None, save=False, last_mlp_rows=rows_g, cols=shared_cols)
        Rt, cap_t = 2 * B + M, 2 * B + cap_M
        t_in = ws.get("t.head_in", (cap_t, D), torch.bfloat16)
        txn = tctx["xn"].view(-1, D)
        ops.gather_rows(txn, D, ix["t_cls"], 2 * B, D, out_bf16=t_in[:2 * B])
        ops.gather_rows(txn, D, patch_rows, M, D, out_bf16=t_in[2 * B:2 * B + M])
        sep = a.ibot_separate_head
        t_logits = ws.get("t.logits_all", (cap_t, K), self.t_head.logit_dtype) if sep else None
        if not sep:
            t_logits = self.t_head.forward(ws, "th", t_in, Rt, cap_t, save=False, segs=[(0, 2 * B), (2 * B, M)],
                                           bn_training=self.teacher_head_training)["logits"]
        else:  # two heads: cls rows through dino_head, masked patch rows through ibot_head
            t_logits[:2 * B].copy_(self.t_head.forward(ws, "th", t_in, 2 * B, 2 * B, save=False,
                                                       bn_training=self.teacher_head_training)["logits"][:2 * B])
            t_logits[2 * B:Rt].copy_(self.t_ihead.forward(ws, "thi", t_in[2 * B:], M, cap_M, save=False,
                                                          bn_training=self.teacher_head_training)["logits"][:M])
        # softmax centering (the default): one pass over the teacher logits leaves the row statistics and the column sums of the center
        # update; the cross-entropy rebuilds the probabilities from the logits (no [rows, K] probability matrix: `teacher_probs()` forms
        # it on demand).  Sinkhorn-Knopp iterates on the matrix and keeps it.  LT_FUSED_CENTERING=0: the three-pass form.
        fused_center = a.center_method == "softmax" and self.fused_centering
        t_probs = None if fused_center else ws.get("t.probs", (cap_t, K), torch.float32)
        t_stats = ws.get("t.stats", (cap_t, 2), torch.float32) if fused_center else None
        if a.center_method == "softmax":
            cs_d = ws.get("t.colsum_dino", (K,), torch.float32)
            cs_i = ws.get("t.colsum_ibot", (K,), torch.float32)
            if fused_center:
                cs_ws = ws.get("t.colsum_ws", (256 * K,), torch.float32)   # per-workgroup column sums (both calls run on this stream)
                ops.softmax_stats_colsum(t_logits[:2 * B], self.dino_center.view(-1), t_stats[:2 * B], cs_d, 2 * B, K, 1.0 / teacher_temp, cs_ws)
                ops.softmax_stats_colsum(t_logits[2 * B:Rt], self.ibot_center.view(-1), t_stats[2 * B:Rt], cs_i, M, K, 1.0 / teacher_temp, cs_ws)
            else:
                ops.softmax_center(t_logits[:2 * B], self.dino_center.view(-1), t_probs[:2 * B], 2 * B, K, 1.0 / teacher_temp)
                ops.softmax_center(t_logits[2 * B:Rt], self.ibot_center.view(-1), t_probs[2 * B:Rt], M, K, 1.0 / teacher_temp)
                ops.colsum_f32(t_logits[:2 * B], cs_d, 2 * B, K)
                ops.colsum_f32(t_logits[2 * B:Rt], cs_i, M, K)
            # dinov2_loss.py:139-145 / :274-282 -- DINO: sum over 2B rows / (2B*world); iBOT: per-rank mean over M / world
            ops.scale_f32(cs_i, 1.0 / max(M, 1))
            hd = hi = None
            if self.world > 1:
                hd = dist.all_reduce(cs_d, async_op=True)
                hi = dist.all_reduce(cs_i, async_op=True)
            self._pending["dino"] = (cs_d, 1.0 / (2 * B), hd)
            self._pending["ibot"] = (cs_i, 1.0, hi)
        else:
            # n_masked_patches over all ranks (dinov2.py:441-449) only scales Q between iterations (see _sinkhorn_joint): this rank's count
            # times the world size stands in for it -- no collective and no host read-back of a device scalar in the step.  Both heads
            # iterate in lockstep: one all-reduce per iteration carries the prototype sums of both
            jobs = [(t_logits[:2 * B], t_probs[:2 * B], 2 * B, float(2 * B * self.world)),
                    (t_logits[2 * B:Rt], t_probs[2 * B:Rt], M, float(max(M, 1) * self.world))]
            if self.sinkhorn_joint:
                self._sinkhorn_joint(jobs, K, teacher_temp, "sk")
            else:
                self._sinkhorn(*jobs[0][:3], K, teacher_temp, jobs[0][3], "skd")
                self._sinkhorn(*jobs[1][:3], K, teacher_temp, jobs[1][3], "ski")

        teacher_done = tstream.record_event()
        torch.cuda.set_stream(main)

        # ---------------- student forward : dinov2.py:474-519
        # stochastic depth draws (student, training): injected (parity tests) or drawn from the host generator
        plan_g = batch.get("drop_plan_global", None) if isinstance(batch, dict) else None
        plan_l = batch.get("drop_plan_local", None) if isinstance(batch, dict) else None
        if plan_g is None:
            plan_g = make_drop_plan(cfg, n_crops, self._drop_gen)
        if plan_l is None and lv is not None:
            plan_l = make_drop_plan(cfg, lv.shape[0], self._drop_gen)
        # joint weight gradients of the two passes: their GEMM operands live side by side (seeded into the workspace before either pass asks)
        self._joint_active = None
        if (self.joint_wgrad and lv is not None and self.side_stream is not None and self.overlap_streams and self.local_bwd_stream is not None
                and self.two_bwd_chains and not self.activation_checkpointing):
            if self._joint is None:
                self._joint = JointWgrad(ws, self.side_stream, ("sg", "sl"))
            hid = cfg.hidden
            if self._joint.seed(cfg.depth, (n_crops * Ng, lv.shape[0] * Nl), D, hid, 2 * hid if cfg.swiglu else hid):
                self._joint_active = self._joint
        # local-crop forward on the side stream, concurrent with the global-crop forward (disjoint activation buffers)
        lstream = self.side_stream if (self.side_stream is not None and self.overlap_streams and lv is not None and self.fwd_local_stream) else None
        sl = None
        if lstream is not None:
            lstream.wait_event(main.record_event())
            with torch.cuda.stream(lstream):
                sl = self.s_vit.forward(ws, "sl", lv, None, save=True, drop_plan=plan_l, checkpoint=self.activation_checkpointing,
                                        last_mlp_rows=rows_l)
                local_done = lstream.record_event()
        sg = self.s_vit.forward(ws, "sg", gv, mask_u8, save=True, drop_plan=plan_g, checkpoint=self.activation_checkpointing,
                                last_mlp_rows=rows_g, cols=shared_cols)
        if lstream is not None:
            main.wait_event(local_done)
        elif lv is not None:
            sl = self.s_vit.forward(ws, "sl", lv, None, save=True, drop_plan=plan_l, checkpoint=self.activation_checkpointing,
                                    last_mlp_rows=rows_l)
        Rl = n_local * B
        Rd = 2 * B + Rl                      # rows of the DINO head: global cls + local cls
        Rs, cap_s = Rd + M, Rd + cap_M       # student row layout [2B cls | Rl local cls | M masked patches]
        s_in = ws.get("s.head_in", (cap_s, D), torch.bfloat16, pad_rows=64)
        sxn = sg["xn"].view(-1, D)
        dxn_g = ws.get("sg.dxn", (2 * B * Ng, D), torch.float32)
        kws = ws.get("koleo.ws", (2 * B * D + 2 * B,), torch.float32)
        knn = ws.get("koleo.nn", (B,), torch.int32)
        # the head phase (cross-entropy -> head backward -> first backbone kernel) is one dependent chain of small kernels: what does not
        # lie on it runs beside it on the side stream -- the KoLeo term (a dozen tiny kernels over the student's cls tokens, which the final
        # norm produced long before the cross-entropy) and the heads' weight gradients.  LT_HEAD_SIDE=0: everything on the main stream
        hside = self.side_stream if (self.head_side and self.side_stream is not None and self.overlap_streams) else None

        dxn_l = ws.get("sl.dxn", (Rl * Nl, D), torch.float32) if sl is not None else None
        rows_g_all = rows_g[0][:2 * B + M] if rows_g is not None else torch.cat([ix["s_cls"][:2 * B], patch_rows[:M]])   # every row of dxn_g a loss writes

        def zero_dxn(buf: Tensor, rows_now: Tensor, key: str) -> None:
            """The upstream-gradient buffer of a pass is zero except at the rows the losses write (cls rows, masked patch rows): after its
            first use only the rows the PREVIOUS step wrote are cleared (an index fill of a few thousand rows instead of 155 / 116 MB of
            HBM writes per step).  A buffer seen for the first time (or re-allocated) is filled whole."""
            prev = self._dxn_rows.get(key)
            if prev is None or prev[0] != buf.data_ptr() or prev[1] != tuple(buf.shape) or os.environ.get("LT_DXN_FULL_ZERO", "0") == "1":
                buf.zero_()
            else:
                if prev[2].is_cuda:
                    # the index tensor was allocated on the main stream and is released below, while this fill may still be queued on the
                    # side stream: tell the caching allocator, or the main stream's next allocation overwrites the indices under the kernel
                    # (seen as a GPU fault with two ranks on one device, tests/test_gpu_ddp.py)
                    prev[2].record_stream(torch.cuda.current_stream())
                buf.index_fill_(0, prev[2], 0.0)
            self._dxn_rows[key] = (buf.data_ptr(), tuple(buf.shape), rows_now)

        def koleo_and_zero() -> None:
            zero_dxn(dxn_g, rows_g_all, "g")
            if dxn_l is not None:
                zero_dxn(dxn_l, ix["l_cls"][:Rl], "l")
            if B > 1:   # weight 0: the kernel only evaluates the term (logged by the reference regardless of its weight)
                kslot = self._loss_slots[3:] if a.koleo_loss_weight != 0.0 else self._loss_slots[4:]
                for c in range(2):  # per global-crop chunk, dinov2.py:377-380
                    ops.koleo_fwd_bwd(sxn[c * B * Ng:], Ng * D, kslot, dxn_g[c * B * Ng:], Ng * D, B, D, a.koleo_loss_weight * gs, kws, knn)

        koleo_done = None
        if hside is None:
            koleo_and_zero()
        else:
            hside.wait_event(main.record_event())
            with torch.cuda.stream(hside):
                koleo_and_zero()
                koleo_done = hside.record_event()

        ops.gather_rows(sxn, D, ix["s_cls"], 2 * B, D, out_bf16=s_in[:2 * B])
        if sl is not None:
            ops.gather_rows(sl["xn"].view(-1, D), D, ix["l_cls"], Rl, D, out_bf16=s_in[2 * B:Rd])
        # (two heads: the patch rows get a buffer of their own -- each head's weight-gradient GEMMs zero the rows up to the next multiple
        # of 64 behind their input, which must not be the other head's rows)
        s_in_i = ws.get("s.head_in_i", (cap_M, D), torch.bfloat16, pad_rows=64) if sep else s_in[Rd:Rs]
        ops.gather_rows(sxn, D, patch_rows, M, D, out_bf16=s_in_i[:M])
        if not sep:
            # BatchNorm segments in the reference's call order: global cls, masked patches (dinov2.py:487-503), local cls (:515)
            sh = self.s_head.forward(ws, "sh", s_in, Rs, cap_s, save=True, segs=[(0, 2 * B), (Rd, M), (2 * B, Rl)])
            shi = None
        else:
            sh = self.s_head.forward(ws, "sh", s_in, Rd, Rd, save=True, segs=[(0, 2 * B), (2 * B, Rl)])
            shi = self.s_ihead.forward(ws, "shi", s_in_i, M, cap_M, save=True)

        # ---------------- losses : dinov2.py:335-387, dinov2_loss.py:117-133,246-268
        r2 = torch.arange(2 * B, dtype=torch.int32)
        ta = torch.cat([r2, torch.arange(B, dtype=torch.int32).repeat(n_local), 2 * B + torch.arange(M, dtype=torch.int32)])
        tb = torch.cat([torch.full((2 * B,), -1, dtype=torch.int32), (B + torch.arange(B, dtype=torch.int32)).repeat(n_local),
                        torch.full((M,), -1, dtype=torch.int32)])
        coef = torch.cat([torch.full((2 * B,), a.dino_loss_weight * 2.0 / terms / (2 * B)), torch.full((Rl,), a.dino_loss_weight / terms / B),
                          a.ibot_loss_weight * mw / n_crops])
        if gs != 1.0:
            coef = coef * gs   # loss / k of a k-batch accumulation window: the slots hold scaled terms, the logs below undo it
        slot = torch.cat([torch.zeros(2 * B, dtype=torch.int32), torch.ones(Rl, dtype=torch.int32), torch.full((M,), 2, dtype=torch.int32)])
        ta, tb, coef, slot = (ops.h2d(t, dev) for t in (ta, tb, coef, slot))
        main.wait_event(teacher_done)
        inv_ts = 1.0 / a.student_temp
        def ce(logits: Tensor, ta_: Tensor, tb_: Tensor, coef_: Tensor, out: Tensor, rows: int, slot_: Tensor) -> None:
            if fused_center:
                ops.ce_fwd_bwd_logits(logits, t_logits, t_stats, self.dino_center.view(-1), self.ibot_center.view(-1), 2 * B, ta_, tb_, coef_, 1.0,
                                      inv_ts, 1.0 / teacher_temp, self._loss_slots, out, rows, K, slot=slot_)
            else:
                ops.ce_fwd_bwd(logits, t_probs, ta_, tb_, coef_, 1.0, inv_ts, self._loss_slots, out, rows, K, slot=slot_)

        if not sep:
            dlogits = ws.get("s.dlogits", (cap_s, K), torch.bfloat16, pad_rows=64)
            ce(sh["logits"], ta, tb, coef, dlogits, Rs, slot)
        else:
            dlogits = ws.get("s.dlogits", (Rd, K), torch.bfloat16, pad_rows=64)
            dlogits_i = ws.get("s.dlogits_i", (cap_M, K), torch.bfloat16, pad_rows=64)
            ce(sh["logits"], ta, tb, coef, dlogits, Rd, slot)
            ce(shi["logits"], ta[Rd:], tb[Rd:], coef[Rd:], dlogits_i, M, slot[Rd:])

        # ---------------- backward
        self._reduce_begin()
        dx_head = self.s_head.backward(ws, sh, dlogits, side=hside)
        self.s_head.finish_weightnorm_grad(hside)
        if koleo_done is not None:
            main.wait_event(koleo_done)
        ops.scatter_add_rows(dx_head[:2 * B], ix["s_cls"], dxn_g, D, 2 * B, D)
        if not sep:
            ops.scatter_add_rows(dx_head[Rd:Rs], patch_rows, dxn_g, D, M, D)
        else:
            dx_ihead = self.s_ihead.backward(ws, shi, dlogits_i, side=hside)
            self.s_ihead.finish_weightnorm_grad(hside)
            ops.scatter_add_rows(dx_ihead[:M], patch_rows, dxn_g, D, M, D)
        if sl is not None:
            ops.scatter_add_rows(dx_head[2 * B:Rd], ix["l_cls"], dxn_l, D, Rl, D)
        self._backward_backbone(sg, dxn_g, sl, dxn_l)

        ls = self._loss_slots
        # slots hold weighted (and, in an accumulation window, 1/k-scaled) terms; report the unweighted terms like the reference's log_dict
        logs = {
            "train_loss/dino_global_loss": ls[0] / (a.dino_loss_weight * gs) if a.dino_loss_weight else ls[0],
            "train_loss/dino_local_loss": ls[1] / (a.dino_loss_weight * gs) if a.dino_loss_weight else ls[1],
            "train_loss/ibot_loss": ls[2] / (a.ibot_loss_weight * gs) if a.ibot_loss_weight else ls[2],
            "train_loss/koleo_loss": ls[3] / (a.koleo_loss_weight * gs) if a.koleo_loss_weight else ls[4],
        }
        self._last_masks = masks
        s_patch_logits = sh["logits"][Rd:Rs] if not sep else shi["logits"][:M]
        self._last = dict(t_cls_logits=t_logits[:2 * B], t_patch_logits=t_logits[2 * B:Rt], t_probs=t_probs[:Rt] if t_probs is not None else None,
                          t_stats=t_stats, teacher_temp=teacher_temp, Rt=Rt,
                          s_cls_logits=sh["logits"][:2 * B], s_local_logits=sh["logits"][2 * B:Rd], s_patch_logits=s_patch_logits,
                          B=B, M=M, Rl=Rl)
        return TrainingStepResult(loss=ls[:4].sum() / gs if gs != 1.0 else ls[:4].sum(), log_dict=logs)

    def teacher_probs(self) -> Tensor:
        """Teacher probabilities f32 [2B + M, K] of the last step ([cls rows, halves swapped | masked patch rows]): the matrix the fused
        centering path never writes, formed on demand with the centers the step used (call before the next step moves them)."""
        L = self._last
        if L.get("t_probs") is not None:
            return L["t_probs"]
        B, M, Rt, K = L["B"], L["M"], L["Rt"], self.method_args.output_dim
        out = torch.empty(Rt, K, dtype=torch.float32, device=self.device)
        ops.softmax_center(L["t_cls_logits"].float(), self.dino_center.view(-1), out[:2 * B], 2 * B, K, 1.0 / L["teacher_temp"])   # (.float(): bf16_logits)
        ops.softmax_center(L["t_patch_logits"].float(), self.ibot_center.view(-1), out[2 * B:], M, K, 1.0 / L["teacher_temp"])
        return out

    def synced_logs(self, res: "TrainingStepResult") -> Dict[str, Tensor]:
        """`train_loss` + `log_dict` averaged over ranks in one coalesced all-reduce (what `Method.training_step` logs with
        `sync_dist=True`, method.py:131-144)."""
        from .parallel import coalesced_mean

        keys = ["train_loss"] + list(res.log_dict)
        vals = coalesced_mean([res.loss] + [torch.as_tensor(res.log_dict[k], device=res.loss.device) for k in res.log_dict])
        return dict(zip(keys, vals))

    def _sinkhorn(self, logits: Tensor, out: Tensor, rows: int, K: int, temp: float, n_total: Any, tag: str) -> None:
        """One head's Sinkhorn-Knopp (see `_sinkhorn_joint`)."""
        self._sinkhorn_joint([(logits, out, rows, n_total)], K, temp, tag)

    def _sinkhorn_joint(self, jobs: List[Tuple[Tensor, Tensor, int, Any]], K: int, temp: float, tag: str) -> None:
        """dinov2_loss.py:84-115 / :188-224 for several (logits, out, rows, n_total) problems that share K and the temperature -- the DINO and
        the iBOT head of one step -- iterated in lockstep: every iteration's prototype sums of ALL problems sit in one [len(jobs) * K] buffer
        and cross the ranks in ONE all-reduce (3 collectives per step instead of 3 per head; the reference issues 1 scalar + 3 vector
        all-reduces per head, dinov2_loss.py:97-106,200-215).  Each problem's arithmetic is untouched: results are bit-identical to
        one-at-a-time calls (tests/test_ddp_gloo.py).
        The initial Q /= sum(Q) is a global scalar that cancels in the first row normalisation, so it is skipped (same value up to fp32
        rounding).  So does the sample count `n_total` (B in the reference): every iteration ends with Q /= B, a factor common to all of Q,
        which the next iteration's Q /= sum_of_rows removes again, and the closing Q *= B undoes the last one -- the result does not depend
        on it beyond rounding, which is why the iBOT call may pass an estimate instead of all-reducing the masked-patch count
        (tests/test_dinov2_method_cpu.py checks it)."""
        nts = [float(n.item()) if isinstance(n, Tensor) else float(n) for _, _, _, n in jobs]
        for logits, out, _, _ in jobs:
            ops.sk_exp(logits, out, 1.0 / temp)
        cs = self.ws.get(tag + ".colsum", (len(jobs) * K,), torch.float32)
        for it in range(3):
            for j, (_, out, rows, _) in enumerate(jobs):
                ops.colsum_f32(out, cs[j * K:(j + 1) * K], rows, K)
            if self.world > 1:
                dist.all_reduce(cs)
            for j, (_, out, rows, _) in enumerate(jobs):
                ops.sk_iter(out, cs[j * K:(j + 1) * K], rows, K, nts[j], nts[j] if it == 2 else 1.0)

    # ------------------------------------------------------------------ optimizer / EMA hooks
    def _gradient_sync(self) -> Optional[GradSync]:
        if self.world == 1:
            return None
        if self._grad_sync is None:
            self._grad_sync = GradSync(self.student.grad)
        return self._grad_sync

    def allreduce_gradients(self) -> None:
        """DDP gradient mean over ranks (C1 in SURVEY.md 2c) on the flat grad buffer (parallel.GradSync): reduces what the
        backward pass has not already started (the small embedding / final-norm tensors, or everything with
        LT_GRAD_OVERLAP=0), waits for all of it on the current stream and scales by 1/world."""
        sync = self._gradient_sync()
        if sync is not None:
            sync.finish()

    def optimizer_step(self) -> Dict[str, float]:
        """on_before_optimizer_step + configure_gradient_clipping + AdamW + CosineWarmupScheduler (dinov2.py:576-639)."""
        a = self.method_args
        k = self.trainer.global_step
        total = self.trainer.estimated_stepping_batches
        wd = cosine_schedule(k, total, a.weight_decay_start, a.weight_decay_end)
        lr_factor = warmup_cosine_lr_factor(k, self.warmup_steps, total, a.min_lr / self.base_lr)
        freeze = (1 if k < a.student_freeze_last_layer_steps else 0) | (2 if k < a.student_freeze_backbone_steps else 0)
        ev0 = None
        if self.comm_events is not None and self.world > 1:
            ev0 = torch.cuda.Event(enable_timing=True)
            ev0.record()
        self.allreduce_gradients()
        if ev0 is not None:
            ev1 = torch.cuda.Event(enable_timing=True)
            ev1.record()
            self.comm_events.append((ev0, ev1))
        if not self._ls_finished:   # an accumulation window that ended on a micro-batch not announced as its last (end of an epoch)
            self.s_vit.finish_layerscale_grads()   # linear in the (now rank-averaged) weight and bias gradients
            self._ls_finished = True
        self._sumsq.zero_()
        ops.sumsq(self.student.grad, self._sumsq)
        self.opt_step += 1
        self._adamw(freeze, lr_factor, wd)
        self.s_head.refresh_weightnorm()
        if self.s_ihead is not self.s_head:
            self.s_ihead.refresh_weightnorm()
        self.s_vit.refresh_padded_weights()
        self.last_grad_norm = self._sumsq  # squared norm, device scalar
        self.trainer.global_step += 1
        return {"weight_decay": wd, "lr_factor": lr_factor}

    def _adamw(self, freeze: int, lr_factor: float, wd: float, lo: int = 0, hi: Optional[int] = None, step: Optional[int] = None) -> None:
        """The fused clip + AdamW launch over elements [lo, hi) of the flat storage (whole tensors: multiples of the 1024-element chunk) at
        Adam step `step` (default: everything at `opt_step`).  A subclass whose extra parameters received no gradient in some steps updates
        them in a launch of their own, at their own step count, as torch.optim.AdamW does per parameter."""
        a = self.method_args
        hi = self.student.data.numel() if hi is None else hi
        assert lo % 1024 == 0 and hi % 1024 == 0
        ops.adamw_flat(self.student.data[lo:hi], self.student.grad[lo:hi], self.exp_avg[lo:hi], self.exp_avg_sq[lo:hi], self.student.bf16[lo:hi],
                       self.student.seg_of_chunk[lo // 1024:hi // 1024], self.seg_lr, self.seg_wd_on, self.seg_frozen, freeze, lr_factor, wd, a.betas[0],
                       a.betas[1], a.eps, self.opt_step if step is None else step, self._sumsq, a.gradient_clip_val)

    def on_train_batch_end(self) -> float:
        """EMA teacher update with momentum evaluated at the already-incremented global_step (dinov2.py:641-660)."""
        a = self.method_args
        m = cosine_schedule(self.trainer.global_step, self.trainer.estimated_stepping_batches, a.momentum_start, a.momentum_end)
        ops.ema_flat(self.teacher.data, self.student.data, self.teacher.bf16, m)
        self.t_head.refresh_weightnorm()
        if self.t_ihead is not self.t_head:
            self.t_ihead.refresh_weightnorm()
        self.t_vit.refresh_padded_weights()
        return m

    def train_step(self, views: List[Tensor], masks: Optional[Dict[str, Tensor]] = None) -> TrainingStepResult:
        res = self.training_step_impl({"views": views}, 0, masks=masks)
        self.optimizer_step()
        self.on_train_batch_end()
        return res
