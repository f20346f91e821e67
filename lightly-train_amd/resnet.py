"""torchvision-style ResNet (bottleneck family: resnet50 / 101 / 152) forward / backward on the HIP ops -- the convolutional
student of BASELINE.json configs[3] ("frozen DINOv3 ViT-L/16 teacher -> torchvision/resnet50 student").

The reference wraps `torchvision.models.resnet50` in `ResNetModelWrapper` (LT/_models/torchvision/resnet.py:21-47:
forward_features = everything up to and including layer4, forward_pool = the model's AdaptiveAvgPool2d) and trains it from
`DistillationV3._forward_student` (LT/_methods/distillationv3/distillationv3.py:324-354).  torchvision itself is not vendored
in the reference tree and not installed in this image: the architecture below restates torchvision's public ResNet v1.5
(stride on the 3x3 convolution, BatchNorm eps 1e-5 / momentum 0.1, kaiming-normal(fan_out) convolutions, no bias) with
state_dict keys identical to `torchvision.models.resnet50().state_dict()`.

MI355X design: activations NHWC bf16 (a [B*H*W, C] matrix), so every convolution is the MFMA GEMM of csrc/gemm.hip -- 1x1
directly, 3x3 / 7x7 on an im2col matrix -- and weights live as [Cout][kh][kw][Cin] in the flat fp32 parameter storage (permuted
to torch's [Cout][Cin][kh][kw] only at state_dict import / export), which makes forward, dgrad and wgrad plain GEMM calls whose
weight gradient lands directly in the flat gradient buffer.  BatchNorm runs in training mode on batch statistics (fp32,
deterministic two-level reductions), fused with ReLU and the residual add (csrc/conv.hip).  No autograd: explicit backward."""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Mapping, Any, Dict, List, Optional, Tuple

import torch
from torch import Tensor

from . import ops
from .params import FlatParams
from .vit import Workspace, _split_k


@dataclass
class ResNetConfig:
    layers: Tuple[int, ...] = (3, 4, 6, 3)
    width: int = 64
    expansion: int = 4
    in_chans: int = 3
    num_classes: int = 1000
    bn_eps: float = 1e-5
    bn_momentum: float = 0.1

    @property
    def feature_dim(self) -> int:
        return self.width * 2 ** (len(self.layers) - 1) * self.expansion


ARCHS: Dict[str, Dict[str, Any]] = {
    "resnet50": dict(layers=(3, 4, 6, 3)),
    "resnet101": dict(layers=(3, 4, 23, 3)),
    "resnet152": dict(layers=(3, 8, 36, 3)),
    "_resnet_test": dict(layers=(1, 1, 1, 1), width=8),   # tiny bottleneck net for parity tests
}


def block_specs(cfg: ResNetConfig) -> List[Dict[str, Any]]:
    """One entry per bottleneck, in torchvision's order: key prefix, channel counts, stride, whether it has a downsample path."""
    out: List[Dict[str, Any]] = []
    inplanes = cfg.width
    for li, n in enumerate(cfg.layers):
        planes = cfg.width * (2 ** li)
        for bi in range(n):
            stride = 2 if (bi == 0 and li > 0) else 1
            down = bi == 0 and (stride != 1 or inplanes != planes * cfg.expansion)
            out.append(dict(prefix=f"layer{li + 1}.{bi}.", inplanes=inplanes, planes=planes, stride=stride, down=down))
            inplanes = planes * cfg.expansion
    return out


def resnet_param_shapes(cfg: ResNetConfig) -> List[Tuple[str, Tuple[int, ...]]]:
    """(torchvision key, torch shape) of every TRAINED parameter in `named_parameters()` order (the classifier `fc.*` is part of
    the exported state_dict but never sees a gradient in distillation: it is held outside the optimizer's flat storage)."""
    out: List[Tuple[str, Tuple[int, ...]]] = [("conv1.weight", (cfg.width, cfg.in_chans, 7, 7)), ("bn1.weight", (cfg.width,)), ("bn1.bias", (cfg.width,))]
    for b in block_specs(cfg):
        p, cin, pl, ex = b["prefix"], b["inplanes"], b["planes"], cfg.expansion
        out += [(p + "conv1.weight", (pl, cin, 1, 1)), (p + "bn1.weight", (pl,)), (p + "bn1.bias", (pl,)),
                (p + "conv2.weight", (pl, pl, 3, 3)), (p + "bn2.weight", (pl,)), (p + "bn2.bias", (pl,)),
                (p + "conv3.weight", (pl * ex, pl, 1, 1)), (p + "bn3.weight", (pl * ex,)), (p + "bn3.bias", (pl * ex,))]
        if b["down"]:
            out += [(p + "downsample.0.weight", (pl * ex, cin, 1, 1)), (p + "downsample.1.weight", (pl * ex,)), (p + "downsample.1.bias", (pl * ex,))]
    return out


def bn_names(cfg: ResNetConfig) -> List[Tuple[str, int]]:
    out = [("bn1", cfg.width)]
    for b in block_specs(cfg):
        p, pl, ex = b["prefix"], b["planes"], cfg.expansion
        out += [(p + "bn1", pl), (p + "bn2", pl), (p + "bn3", pl * ex)]
        if b["down"]:
            out.append((p + "downsample.1", pl * ex))
    return out


def init_resnet_state(cfg: ResNetConfig, generator: Optional[torch.Generator] = None) -> Dict[str, Tensor]:
    """torchvision's initialisation (resnet.py `ResNet.__init__`): kaiming_normal_(mode="fan_out", nonlinearity="relu") for
    convolutions, BatchNorm weight 1 / bias 0 and fresh running statistics, nn.Linear default for `fc`."""
    sd: Dict[str, Tensor] = {}
    for name, shape in resnet_param_shapes(cfg):
        if len(shape) == 4:
            fan_out = shape[0] * shape[2] * shape[3]
            sd[name] = torch.empty(shape).normal_(0.0, math.sqrt(2.0 / fan_out), generator=generator)
        elif name.endswith(".weight"):
            sd[name] = torch.ones(shape)
        else:
            sd[name] = torch.zeros(shape)
    for bn, c in bn_names(cfg):
        sd[bn + ".running_mean"] = torch.zeros(c)
        sd[bn + ".running_var"] = torch.ones(c)
        sd[bn + ".num_batches_tracked"] = torch.zeros((), dtype=torch.long)
    bound = 1.0 / math.sqrt(cfg.feature_dim)
    sd["fc.weight"] = torch.empty(cfg.num_classes, cfg.feature_dim).uniform_(-bound, bound, generator=generator)
    sd["fc.bias"] = torch.empty(cfg.num_classes).uniform_(-bound, bound, generator=generator)
    return sd


def state_dict_order(cfg: ResNetConfig) -> List[str]:
    """Keys of `torchvision.models.resnet50().state_dict()` in order (parameters and buffers interleaved per module)."""
    def bn(p: str) -> List[str]:
        return [p + ".weight", p + ".bias", p + ".running_mean", p + ".running_var", p + ".num_batches_tracked"]

    keys = ["conv1.weight"] + bn("bn1")
    for b in block_specs(cfg):
        p = b["prefix"]
        keys += [p + "conv1.weight"] + bn(p + "bn1") + [p + "conv2.weight"] + bn(p + "bn2") + [p + "conv3.weight"] + bn(p + "bn3")
        if b["down"]:
            keys += [p + "downsample.0.weight"] + bn(p + "downsample.1")
    return keys + ["fc.weight", "fc.bias"]


def to_flat_layout(name: str, t: Tensor) -> Tensor:
    """torch [Cout, Cin, kh, kw] -> the engine's [Cout, kh, kw, Cin] for the 1x1 / 3x3 convolutions (the 7x7 stem stays in torch
    layout: its im2col matrix is built in weight.flatten(1) order)."""
    if t.dim() == 4 and name != "conv1.weight":
        return t.permute(0, 2, 3, 1).contiguous()
    return t


def from_flat_layout(name: str, t: Tensor) -> Tensor:
    if t.dim() == 4 and name != "conv1.weight":
        return t.permute(0, 3, 1, 2).contiguous()
    return t


def flat_named(cfg: ResNetConfig, sd: Dict[str, Tensor], prefix: str = "") -> List[Tuple[str, Tensor]]:
    return [(prefix + n, to_flat_layout(n, sd[n].float())) for n, _ in resnet_param_shapes(cfg)]


def default_bn_sync():
    """The all-reduce SyncBatchNorm needs, when this process is one of several ranks; None otherwise."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return lambda t: dist.all_reduce(t)
    return None


class ResNetEngine:
    """Runs one ResNet whose trained parameters live in a FlatParams under `prefix` (engine layout, see `to_flat_layout`)."""

    def __init__(self, cfg: ResNetConfig, params: FlatParams, prefix: str = "", buffers: Optional[Dict[str, Tensor]] = None) -> None:
        self.cfg, self.P, self.prefix, self.dev = cfg, params, prefix, params.device
        self.blocks = block_specs(cfg)
        self.buffers: Dict[str, Tensor] = {}
        for bn, c in bn_names(cfg):
            for k, init in ((".running_mean", torch.zeros(c)), (".running_var", torch.ones(c)), (".num_batches_tracked", torch.zeros((), dtype=torch.long))):
                src = buffers[bn + k] if buffers is not None and (bn + k) in buffers else init
                # the batch counters live on the host: `+= 1` per layer per step is then not a kernel launch (53 of them per ResNet-50 step)
                self.buffers[bn + k] = src.detach().clone().to("cpu" if k == ".num_batches_tracked" else self.dev)
        self.act_dtype = torch.bfloat16   # storage type of activations / activation gradients (the HIP ops only accept bf16)
        # SyncBatchNorm: Lightning replaces every BatchNorm layer when sync_batchnorm=True, which the reference sets whenever the accelerator is a
        # GPU (LT/_commands/train_helpers.py:223,335-342).  None = per-process statistics (one rank); else a callable that adds a device tensor
        # of doubles over the ranks in place -- set by default as soon as a process group with more than one rank exists.
        self.bn_sync = default_bn_sync()
        self.kreal = cfg.in_chans * 49
        self.kpad = (self.kreal + 7) // 8 * 8
        self.w_stem = torch.zeros(cfg.width, self.kpad, dtype=torch.bfloat16, device=self.dev)
        self.refresh_padded_weights()

    def refresh_padded_weights(self) -> None:
        """bf16 [Cout, 152] copy of the 7x7 stem matrix (3*49 = 147 is not a multiple of 8) after the fp32 weights changed."""
        ops.cast_pad_rows(self.w("conv1.weight").view(self.cfg.width, -1), self.w_stem, self.cfg.width, self.kreal, self.kpad)

    def w(self, n: str) -> Tensor:
        return self.P.p[self.prefix + n]

    def wb(self, n: str) -> Tensor:
        return self.P.b[self.prefix + n]

    def gw(self, n: str) -> Tensor:
        return self.P.g[self.prefix + n]

    # ---- one conv (+ BatchNorm) unit ---------------------------------------------------------------
    def _bn_fwd(self, ws: Workspace, bn: str, x: Tensor, y: Tensor, rows: int, C: int, relu: bool, resid: Optional[Tensor], train: bool,
                tag: str) -> Tuple[Tensor, Tensor]:
        mean = ws.get(f"{tag}.{bn}.mean", (C,), torch.float32)
        rstd = ws.get(f"{tag}.{bn}.rstd", (C,), torch.float32)
        scratch = ws.get("bn.ws", (ops.batchnorm_ws_floats(max(2048, self.cfg.feature_dim)),), torch.float32)
        if train:
            ops.batchnorm_fwd(x, self.w(bn + ".weight"), self.w(bn + ".bias"), y, mean, rstd, rows, C, scratch, resid=resid,
                              running_mean=self.buffers[bn + ".running_mean"], running_var=self.buffers[bn + ".running_var"],
                              eps=self.cfg.bn_eps, momentum=self.cfg.bn_momentum, relu=relu, sync=self.bn_sync)
            self.buffers[bn + ".num_batches_tracked"] += 1
        else:   # eval mode: the running estimates stand in for the batch statistics (plumbing: two [C] vectors)
            mean.copy_(self.buffers[bn + ".running_mean"])
            torch.rsqrt(self.buffers[bn + ".running_var"] + self.cfg.bn_eps, out=rstd)
            ops.batchnorm_apply(x, mean, rstd, self.w(bn + ".weight"), self.w(bn + ".bias"), y, rows, C, resid=resid, relu=relu)
        return mean, rstd

    def forward(self, ws: Workspace, tag: str, img: Tensor, save: bool = True, train: bool = True) -> Dict[str, Any]:
        """img f32 [B, 3, H, W] -> ctx with "feat" bf16 [B*h*w, feature_dim] (NHWC rows of the layer4 output), "h", "w"."""
        cfg = self.cfg
        B, Cin, H, W = img.shape
        Wd = cfg.width
        bf = self.act_dtype

        def rows_pad(r: int) -> int:
            return (r + 63) // 64 * 64   # whole 64-row k-tiles for the weight-gradient GEMMs; pad rows stay zero

        # ---- stem: 7x7/2 conv (im2col in weight.flatten(1) order) + BN + ReLU + 3x3/2 max-pool
        H1, W1 = ops.conv_out_size(H, 7, 2, 3), ops.conv_out_size(W, 7, 2, 3)
        r1 = B * H1 * W1
        cols0 = ws.get(f"{tag}.stem.cols", (rows_pad(r1), self.kpad), bf, zero=True)
        ops.im2col_nchw_f32(img.contiguous(), cols0[:r1], 7, 7, 2, 3)
        c0 = ws.get(f"{tag}.stem.c", (rows_pad(r1), Wd), bf, zero=True)
        ops.gemm(cols0, self.w_stem, c0, M=r1, N=Wd, K=self.kpad, epilogue=ops.EPI_BF16)
        a0 = ws.get(f"{tag}.stem.a", (r1, Wd), bf)
        m0, s0 = self._bn_fwd(ws, "bn1", c0, a0, r1, Wd, True, None, train, tag)
        H2, W2 = ops.conv_out_size(H1, 3, 2, 1), ops.conv_out_size(W1, 3, 2, 1)
        r2 = B * H2 * W2
        x = ws.get(f"{tag}.stem.pool", (rows_pad(r2), Wd), bf, zero=True)
        pidx = ws.get(f"{tag}.stem.pidx", (r2, Wd), torch.uint8)
        ops.maxpool3x3s2_fwd(a0, x, pidx, B, H1, W1, Wd)
        ctx: Dict[str, Any] = dict(B=B, tag=tag, stem=dict(cols=cols0, c=c0, a=a0, mean=m0, rstd=s0, pidx=pidx, H1=H1, W1=W1, r1=r1, H2=H2, W2=W2, r2=r2),
                                   blocks=[])
        h, w_ = H2, W2
        for bi, b in enumerate(self.blocks):
            p, cin, pl, st = b["prefix"], b["inplanes"], b["planes"], b["stride"]
            cout = pl * cfg.expansion
            t = f"{tag}.{p}"
            rin = B * h * w_
            ho, wo = ops.conv_out_size(h, 3, st, 1), ops.conv_out_size(w_, 3, st, 1)
            rout = B * ho * wo
            c1 = ws.get(t + "c1", (rows_pad(rin), pl), bf, zero=True)
            ops.gemm(x, self.wb(p + "conv1.weight").view(pl, cin), c1, M=rin, N=pl, K=cin, epilogue=ops.EPI_BF16)
            a1 = ws.get(t + "a1", (rin, pl), bf)
            m1, s1 = self._bn_fwd(ws, p + "bn1", c1, a1, rin, pl, True, None, train, tag)
            cols2 = ws.get(t + "cols2", (rows_pad(rout), 9 * pl), bf, zero=True)
            ops.im2col_nhwc(a1, cols2[:rout], B, h, w_, pl, 3, 3, st, 1)
            c2 = ws.get(t + "c2", (rows_pad(rout), pl), bf, zero=True)
            ops.gemm(cols2, self.wb(p + "conv2.weight").view(pl, 9 * pl), c2, M=rout, N=pl, K=9 * pl, epilogue=ops.EPI_BF16)
            a2 = ws.get(t + "a2", (rows_pad(rout), pl), bf, zero=True)
            m2, s2 = self._bn_fwd(ws, p + "bn2", c2, a2, rout, pl, True, None, train, tag)
            c3 = ws.get(t + "c3", (rows_pad(rout), cout), bf, zero=True)
            ops.gemm(a2, self.wb(p + "conv3.weight").view(cout, pl), c3, M=rout, N=cout, K=pl, epilogue=ops.EPI_BF16)
            rec: Dict[str, Any] = dict(b, x=x, c1=c1, a1=a1, m1=m1, s1=s1, cols2=cols2, c2=c2, a2=a2, m2=m2, s2=s2, c3=c3, h=h, w=w_, ho=ho, wo=wo,
                                       rin=rin, rout=rout)
            if b["down"]:
                if st != 1:
                    xs = ws.get(t + "xs", (rows_pad(rout), cin), bf, zero=True)
                    ops.im2col_nhwc(x, xs[:rout], B, h, w_, cin, 1, 1, st, 0)
                else:
                    xs = x
                cd = ws.get(t + "cd", (rows_pad(rout), cout), bf, zero=True)
                ops.gemm(xs, self.wb(p + "downsample.0.weight").view(cout, cin), cd, M=rout, N=cout, K=cin, epilogue=ops.EPI_BF16)
                ident = ws.get(t + "ident", (rout, cout), bf)
                md, sd_ = self._bn_fwd(ws, p + "downsample.1", cd, ident, rout, cout, False, None, train, tag)
                rec.update(xs=xs, cd=cd, md=md, sd=sd_)
            else:
                ident = x
            out = ws.get(t + "out", (rows_pad(rout), cout), bf, zero=True)
            m3, s3 = self._bn_fwd(ws, p + "bn3", c3, out, rout, cout, True, ident[:rout], train, tag)
            rec.update(m3=m3, s3=s3, out=out)
            ctx["blocks"].append(rec)
            x, h, w_ = out, ho, wo
        ctx.update(feat=x, h=h, w=w_, rows=B * h * w_)
        return ctx

    def backward(self, ws: Workspace, ctx: Dict[str, Any], dfeat: Tensor, side: Optional["torch.cuda.Stream"] = None) -> None:
        """dfeat bf16 [B*h*w, feature_dim] = dL/d(layer4 output).  Accumulates into the FlatParams gradient views.
        `side`: optional second HIP stream for the weight-gradient GEMMs (they feed nothing but the optimizer)."""
        cfg = self.cfg
        B, tag = ctx["B"], ctx["tag"]
        bf = self.act_dtype
        slab = ws.get("wgrad.slabs", (32 * 1024 * 1024,), torch.float32)
        bnws = ws.get("bn.ws", (ops.batchnorm_ws_floats(max(2048, cfg.feature_dim)),), torch.float32)
        main = torch.cuda.current_stream() if side is not None else None

        def wgrad(dy: Tensor, xin: Tensor, gview: Tensor, n_out: int, k_in: int, rows: int) -> None:
            tiles = ((n_out + 127) // 128) * ((k_in + 127) // 128)
            kpad = (rows + 63) // 64 * 64
            if kpad > dy.shape[0] or kpad > xin.shape[0]:
                kpad = rows

            def run() -> None:
                ops.gemm(dy, xin, gview, M=n_out, N=k_in, K=kpad, trans_a=True, trans_b=True, epilogue=ops.EPI_F32_ACCUM,
                         split_k=_split_k(tiles, kpad), lda=n_out, ldb=k_in, ldc=k_in, workspace=slab)

            if side is None:
                run()
            else:
                side.wait_event(main.record_event())
                with torch.cuda.stream(side):
                    run()

        def bn_bwd(bn: str, dy: Tensor, y: Optional[Tensor], x: Tensor, mean: Tensor, rstd: Tensor, dz: Optional[Tensor], dx: Tensor, rows: int, C: int) -> None:
            ops.batchnorm_bwd(dy, x, self.w(bn + ".weight"), mean, rstd, dx, rows, C, bnws, y=y, dz=dz, dgamma=self.gw(bn + ".weight"),
                              dbeta=self.gw(bn + ".bias"), sync=self.bn_sync)

        def scratch(name: str, shape: Tuple[int, int]) -> Tensor:
            """View of a grow-only 1-D scratch buffer (main-stream temporaries whose shape changes from block to block)."""
            n = shape[0] * shape[1]
            buf = ws.bufs.get(name)
            if buf is None or buf.numel() < n or buf.dtype != bf:
                buf = torch.empty(n, dtype=bf, device=self.dev)
                ws.bufs[name] = buf
            return buf[:n].view(shape)

        d_out = dfeat
        for rec in reversed(ctx["blocks"]):
            p, cin, pl, st = rec["prefix"], rec["inplanes"], rec["planes"], rec["stride"]
            cout = pl * cfg.expansion
            t = f"{tag}.{p}"
            rin, rout, h, w_ = rec["rin"], rec["rout"], rec["h"], rec["w"]
            rp_in, rp_out = rec["c1"].shape[0], rec["c3"].shape[0]
            # ---- bn3 (+ ReLU of the block output, + the residual add: dz3 also flows down the identity path)
            dz3 = ws.get(t + "dz3", (rp_out, cout), bf, zero=True)
            dc3 = ws.get(t + "dc3", (rp_out, cout), bf, zero=True)
            bn_bwd(p + "bn3", d_out, rec["out"], rec["c3"], rec["m3"], rec["s3"], dz3, dc3, rout, cout)
            wgrad(dc3, rec["a2"], self.gw(p + "conv3.weight").view(cout, pl), cout, pl, rout)
            da2 = scratch(f"{tag}.da", (rp_out, pl))
            ops.gemm(dc3, self.wb(p + "conv3.weight").view(cout, pl), da2, M=rout, N=pl, K=cout, trans_b=True, epilogue=ops.EPI_BF16)
            # ---- bn2 + ReLU, 3x3 convolution
            dzs = scratch(f"{tag}.dzs", (rp_out, pl))
            dc2 = ws.get(t + "dc2", (rp_out, pl), bf, zero=True)
            bn_bwd(p + "bn2", da2, rec["a2"], rec["c2"], rec["m2"], rec["s2"], dzs, dc2, rout, pl)
            wgrad(dc2, rec["cols2"], self.gw(p + "conv2.weight").view(pl, 9 * pl), pl, 9 * pl, rout)
            dcols = scratch(f"{tag}.dcols", (rout, 9 * pl))
            ops.gemm(dc2, self.wb(p + "conv2.weight").view(pl, 9 * pl), dcols, M=rout, N=9 * pl, K=pl, trans_b=True, epilogue=ops.EPI_BF16)
            da1 = scratch(f"{tag}.da1", (rin, pl))
            ops.col2im_nhwc(dcols, da1, B, h, w_, pl, 3, 3, st, 1)
            # ---- bn1 + ReLU, first 1x1 convolution
            dzs1 = scratch(f"{tag}.dzs1", (rin, pl))
            dc1 = ws.get(t + "dc1", (rp_in, pl), bf, zero=True)
            bn_bwd(p + "bn1", da1, rec["a1"], rec["c1"], rec["m1"], rec["s1"], dzs1, dc1, rin, pl)
            wgrad(dc1, rec["x"], self.gw(p + "conv1.weight").view(pl, cin), pl, cin, rin)
            dx1 = scratch(f"{tag}.dx1", (rin, cin))
            ops.gemm(dc1, self.wb(p + "conv1.weight").view(pl, cin), dx1, M=rin, N=cin, K=pl, trans_b=True, epilogue=ops.EPI_BF16)
            # ---- identity path
            dx = ws.get(t + "dx", (rin, cin), bf)
            if rec["down"]:
                dcd = ws.get(t + "dcd", (rp_out, cout), bf, zero=True)
                bn_bwd(p + "downsample.1", dz3, None, rec["cd"], rec["md"], rec["sd"], None, dcd, rout, cout)
                wgrad(dcd, rec["xs"], self.gw(p + "downsample.0.weight").view(cout, cin), cout, cin, rout)
                dxs = scratch(f"{tag}.dxs", (rout, cin))
                ops.gemm(dcd, self.wb(p + "downsample.0.weight").view(cout, cin), dxs, M=rout, N=cin, K=cout, trans_b=True, epilogue=ops.EPI_BF16)
                if st != 1:
                    ops.col2im_nhwc(dxs, dx, B, h, w_, cin, 1, 1, st, 0, add=dx1)
                else:
                    ops.add_bf16(dx1, dxs, dx)
            else:
                ops.add_bf16(dx1, dz3[:rin], dx)
            d_out = dx
        # ---- stem: max-pool -> BN + ReLU -> 7x7 convolution (no input gradient)
        s = ctx["stem"]
        Wd = cfg.width
        da0 = ws.get(f"{tag}.stem.da", (s["r1"], Wd), bf)
        ops.maxpool3x3s2_bwd(d_out, s["pidx"], da0, B, s["H1"], s["W1"], Wd)
        dz0 = ws.get(f"{tag}.stem.dz", (s["r1"], Wd), bf)
        dc0 = ws.get(f"{tag}.stem.dc", (s["c"].shape[0], Wd), bf, zero=True)
        bn_bwd("bn1", da0, s["a"], s["c"], s["mean"], s["rstd"], dz0, dc0, s["r1"], Wd)
        gpad = ws.get(f"{tag}.stem.gw", (Wd, self.kpad), torch.float32)
        gpad.zero_()
        wgrad(dc0, s["cols"], gpad, Wd, self.kpad, s["r1"])

        def fold() -> None:
            ops.unpad_accumulate(gpad, self.gw("conv1.weight").view(Wd, -1), Wd, self.kreal, self.kpad)

        if side is None:
            fold()
        else:
            with torch.cuda.stream(side):
                fold()

    # ---- state dict in torchvision's layout -------------------------------------------------------
    def state_dict(self, extra: Optional[Dict[str, Tensor]] = None) -> Dict[str, Tensor]:
        out: Dict[str, Tensor] = {}
        for k in state_dict_order(self.cfg):
            if (self.prefix + k) in self.P.p:
                out[k] = from_flat_layout(k, self.P.p[self.prefix + k].detach().clone())
            elif k in self.buffers:
                out[k] = self.buffers[k].detach().clone()
            elif extra is not None and k in extra:
                out[k] = extra[k].detach().clone()
        return out

    def load_state_dict(self, sd: Dict[str, Tensor]) -> None:
        for n, _ in resnet_param_shapes(self.cfg):
            self.P.p[self.prefix + n].copy_(to_flat_layout(n, sd[n].float()).to(self.dev))
            self.P.b[self.prefix + n].copy_(self.P.p[self.prefix + n])
        self.load_buffers(sd)
        self.refresh_padded_weights()

    def load_buffers(self, sd: Mapping[str, Tensor]) -> None:
        """BatchNorm running estimates / batch counters present in `sd` (torchvision names)."""
        for k in self.buffers:
            if k in sd:
                self.buffers[k].copy_(sd[k].to(self.buffers[k].device))
