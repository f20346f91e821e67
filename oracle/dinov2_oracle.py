"""TEST INFRASTRUCTURE ONLY -- CPU restatement (plain torch fp32 / fp64) of the
reference's DINOv2 training step.  The product path (lightly-train_amd/) must never
import this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.

Parity status: PINNED.  Every function below is checked (tests/test_oracle_pin.py)
  * against the reference's own known-answer tests (DINOLoss 1.5565, IBOTPatchLoss
    0.4057, center 0.2, EMA [[2.5,3.5],[4.5,5.5]], LR-schedule endpoints), and
  * against outputs of the reference's own code run in the build container
    (oracle/ref_harness.py imports /root/reference directly; oracle/make_golden.py
    writes the fixtures under tests/golden/).
Exceptions (un-vendored LightlySSL code): KoLeoLoss, cosine_schedule,
CosineWarmupScheduler and update_param_groups are restated from the published
LightlySSL algorithm (requirement `lightly>=1.5.26`, not present in /root/reference);
KoLeo and the EMA/WD cosine values are "parity unpinned" against the real package.

All arithmetic is floating point (fp32 by default, fp64 selectable) -- this is the
"torch fp32 reference" for the floating-point kernels.

Reference citations are relative to /root/reference/src/lightly_train (LT/).
"""
from __future__ import annotations

import math
import random
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F
from torch import Tensor

# --------------------------------------------------------------------------------------
# Un-vendored LightlySSL helpers (restated from the published algorithm)
# --------------------------------------------------------------------------------------


def cosine_schedule(step: int, max_steps: int, start_value: float, end_value: float,
                    period: Optional[int] = None) -> float:
    """lightly.utils.scheduler.cosine_schedule (call sites LT/_methods/dinov2/dinov2.py:602-607,648-653)."""
    if step < 0:
        raise ValueError("step must be >= 0")
    if max_steps < 1:
        raise ValueError("max_steps must be >= 1")
    if period is None and step > max_steps:
        raise ValueError("step > max_steps")
    if period is not None:
        return end_value - (end_value - start_value) * (math.cos(2 * math.pi * step / period) + 1) / 2
    if max_steps == 1 or step == max_steps:
        return end_value
    return end_value - (end_value - start_value) * (math.cos(math.pi * step / (max_steps - 1)) + 1) / 2


def cosine_warmup_factor(epoch: int, warmup_epochs: int, max_epochs: int, end_value: float) -> float:
    """LR multiplier of lightly.utils.scheduler.CosineWarmupScheduler (LT/_methods/dinov2/dinov2.py:576-586)."""
    if epoch < warmup_epochs:
        return (epoch + 1) / warmup_epochs
    if max_epochs is not None and epoch >= max_epochs:
        return end_value
    return cosine_schedule(epoch - warmup_epochs, max_epochs - warmup_epochs, 1.0, end_value)


class CosineWarmupScheduler(torch.optim.lr_scheduler.LambdaLR):
    def __init__(self, optimizer, warmup_epochs: int, max_epochs: int, last_epoch: int = -1,
                 start_value: float = 1.0, end_value: float = 0.001, **_: Any) -> None:
        self.warmup_epochs = warmup_epochs
        self.max_epochs = max_epochs
        self.end_value = end_value
        super().__init__(optimizer, lr_lambda=self.scale_lr, last_epoch=last_epoch)

    def scale_lr(self, epoch: int) -> float:
        return cosine_warmup_factor(epoch, self.warmup_epochs, self.max_epochs, self.end_value)


def update_param_groups(optimizer, updates: List[Dict[str, Any]]) -> None:
    """lightly.utils.optim.update_param_groups: match groups by "name", copy the other keys."""
    by_name = {g["name"]: g for g in optimizer.param_groups}
    for upd in updates:
        g = by_name.get(upd["name"])
        if g is None:
            continue
        for k, v in upd.items():
            if k != "name":
                g[k] = v


class KoLeoLoss(torch.nn.Module):
    """lightly.loss.KoLeoLoss(p=2, eps=1e-8) (call site LT/_methods/dinov2/dinov2.py:257,377-380)."""

    def __init__(self, p: float = 2, eps: float = 1e-8) -> None:
        super().__init__()
        self.p = p
        self.eps = eps

    def forward(self, x: Tensor) -> Tensor:
        return koleo_loss(x, self.p, self.eps)


def koleo_loss(x: Tensor, p: float = 2, eps: float = 1e-8) -> Tensor:
    x = F.normalize(x, p=2, dim=-1, eps=eps)
    cos = x @ x.t()
    cos = cos.clone()
    cos.fill_diagonal_(-2.0)
    nn_idx = cos.argmax(dim=1)
    # torch.nn.PairwiseDistance(p, eps): || a - b + eps ||_p
    dist = torch.linalg.vector_norm(x - x[nn_idx] + eps, ord=p, dim=-1)
    return -torch.log(dist + eps).mean()


# --------------------------------------------------------------------------------------
# Schedules / masks (host logic)  LT/_methods/dinov2/scheduler.py:13-34, utils.py:41-152
# --------------------------------------------------------------------------------------


def linear_warmup_schedule(step: int, warmup_steps: int, start_value: float, end_value: float) -> float:
    if step < warmup_steps:
        return start_value + step / warmup_steps * (end_value - start_value)
    return end_value


class BlockMaskSampler:
    """Restates MaskingGenerator (LT/_methods/dinov2/utils.py:41-113); RNG = python `random`."""

    def __init__(self, grid: Tuple[int, int], max_num_patches: int, min_num_patches: int = 4,
                 min_aspect: float = 0.3) -> None:
        self.gh, self.gw = grid
        self.num_patches = self.gh * self.gw
        self.min_n = min_num_patches
        self.max_n = max_num_patches
        self.log_ar = (math.log(min_aspect), math.log(1 / min_aspect))

    def _place(self, mask: np.ndarray, budget: int) -> int:
        added = 0
        for _ in range(10):
            area = random.uniform(self.min_n, budget)
            ar = math.exp(random.uniform(*self.log_ar))
            bh = int(round(math.sqrt(area * ar)))
            bw = int(round(math.sqrt(area / ar)))
            if bw < self.gw and bh < self.gh:
                top = random.randint(0, self.gh - bh)
                left = random.randint(0, self.gw - bw)
                already = int(mask[top:top + bh, left:left + bw].sum())
                if 0 < bh * bw - already <= budget:
                    sub = mask[top:top + bh, left:left + bw]
                    added += int((~sub).sum())
                    sub[...] = True
            if added > 0:
                break
        return added

    def sample(self, n_target: int) -> np.ndarray:
        mask = np.zeros((self.gh, self.gw), dtype=bool)
        count = 0
        while count < n_target:
            budget = min(n_target - count, self.max_n)
            d = self._place(mask, budget)
            if d == 0:
                break
            count += d
        return mask


def make_collated_masks(ratio_min: float, ratio_max: float, n_masked_crops: int, n_crops: int,
                        sampler: BlockMaskSampler) -> Dict[str, Tensor]:
    """Restates create_collated_masks (LT/_methods/dinov2/utils.py:116-152)."""
    n_p = sampler.num_patches
    edges = np.linspace(ratio_min, ratio_max, n_masked_crops + 1)
    masks: List[Tensor] = []
    for i in range(n_masked_crops):
        masks.append(torch.from_numpy(sampler.sample(int(n_p * random.uniform(edges[i], edges[i + 1])))))
    for _ in range(n_masked_crops, n_crops):
        masks.append(torch.from_numpy(sampler.sample(0)))
    random.shuffle(masks)
    collated = torch.stack(masks).flatten(1)
    idx = collated.flatten().nonzero().flatten()
    per_crop = 1.0 / collated.sum(-1).clamp(min=1.0)
    weight = per_crop.unsqueeze(-1).expand_as(collated)[collated]
    return {"collated_masks": collated, "mask_indices_list": idx, "masks_weight": weight}


# --------------------------------------------------------------------------------------
# ViT backbone (functional, on a reference-keyed state dict)
# LT/_models/dinov2_vit/dinov2_vit_src/models/vision_transformer.py:251-384, layers/*.py
# --------------------------------------------------------------------------------------


def pos_embed_for_grid(pos_embed: Tensor, gh: int, gw: int, interpolate_offset: float = 0.1,
                       antialias: bool = False) -> Tensor:
    """[1, 1+M*M, D] -> [1, 1+gh*gw, D] (vision_transformer.py:251-305)."""
    n_native = pos_embed.shape[1] - 1
    m = int(math.sqrt(n_native))
    if gh * gw == n_native and gh == gw:
        return pos_embed
    d = pos_embed.shape[-1]
    pe = pos_embed.float()
    cls_pe, patch_pe = pe[:, :1], pe[:, 1:]
    grid = patch_pe.reshape(1, m, m, d).permute(0, 3, 1, 2)
    if interpolate_offset:
        kw: Dict[str, Any] = {"scale_factor": (float(gh + interpolate_offset) / m, float(gw + interpolate_offset) / m)}
    else:
        kw = {"size": (gh, gw)}
    grid = F.interpolate(grid, mode="bicubic", antialias=antialias, **kw)
    assert grid.shape[-2:] == (gh, gw)
    return torch.cat([cls_pe, grid.permute(0, 2, 3, 1).reshape(1, gh * gw, d)], dim=1).to(pos_embed.dtype)


def block_drop_rates(cfg: Dict[str, Any]) -> List[float]:
    """vision_transformer.py:150-157: uniform rate or linspace(0, rate, depth)."""
    rate, depth = float(cfg.get("drop_path_rate", 0.0)), cfg["depth"]
    if cfg.get("drop_path_uniform", False):
        return [rate] * depth
    return [x.item() for x in torch.linspace(0, rate, depth)]


def vit_forward(p: Dict[str, Tensor], x: Tensor, cfg: Dict[str, Any], masks: Optional[Tensor] = None,
                capture: Optional[Dict[str, Tensor]] = None, drop: Any = None) -> Dict[str, Tensor]:
    """x [B,C,H,W] -> {"cls":[B,D], "patch":[B,n_p,D], "prenorm":[B,N,D]}.

    drop: None (eval / teacher), "torch" (training: draw like the reference from torch's global RNG,
    layers/block.py:118-141 + drop_path.py:16-28), or a list (one entry per residual branch, attn then ffn per block)
    of None | ("subset", brange LongTensor) | ("persample", scale FloatTensor[B]) to inject the draws."""
    ps, heads, depth = cfg["patch_size"], cfg["num_heads"], cfg["depth"]
    B, _, H, W = x.shape
    nh, nw = math.ceil(H / ps) * ps, math.ceil(W / ps) * ps
    if (nh, nw) != (H, W):  # layers/patch_embed.py:90-99
        x = F.interpolate(x, size=(nh, nw), mode="bicubic", align_corners=False)
    t = F.conv2d(x, p["patch_embed.proj.weight"], p["patch_embed.proj.bias"], stride=ps)
    gh, gw = t.shape[2], t.shape[3]
    t = t.flatten(2).transpose(1, 2)
    if masks is not None:
        t = torch.where(masks.unsqueeze(-1), p["mask_token"].to(t.dtype).unsqueeze(0), t)
    t = torch.cat([p["cls_token"].expand(B, -1, -1), t], dim=1)
    t = t + pos_embed_for_grid(p["pos_embed"], gh, gw, cfg.get("interpolate_offset", 0.1),
                               cfg.get("interpolate_antialias", False))
    if "register_tokens" in p:
        t = torch.cat([t[:, :1], p["register_tokens"].expand(B, -1, -1), t[:, 1:]], dim=1)
    if capture is not None:
        capture["tokens"] = t
    D = t.shape[-1]
    dh = D // heads
    has_ls = "blocks.0.ls1.gamma" in p
    rates = block_drop_rates(cfg)
    used_draws: List[Any] = []

    def attn_branch(z: Tensor, pre: str) -> Tensor:
        b_ = z.shape[0]
        y = F.layer_norm(z, (D,), p[pre + "norm1.weight"], p[pre + "norm1.bias"], 1e-6)
        qkv = F.linear(y, p[pre + "attn.qkv.weight"], p[pre + "attn.qkv.bias"])
        qkv = qkv.reshape(b_, -1, 3, heads, dh).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0] * dh ** -0.5, qkv[1], qkv[2]
        a = (q @ k.transpose(-2, -1)).softmax(dim=-1)
        y = (a @ v).transpose(1, 2).reshape(b_, -1, D)
        y = F.linear(y, p[pre + "attn.proj.weight"], p[pre + "attn.proj.bias"])
        return y * p[pre + "ls1.gamma"] if has_ls else y

    def ffn_branch(z: Tensor, pre: str) -> Tensor:
        y = F.layer_norm(z, (D,), p[pre + "norm2.weight"], p[pre + "norm2.bias"], 1e-6)
        if pre + "mlp.w12.weight" in p:   # SwiGLUFFNFused, layers/swiglu_ffn.py:31-35
            x1, x2 = F.linear(y, p[pre + "mlp.w12.weight"], p[pre + "mlp.w12.bias"]).chunk(2, dim=-1)
            y = F.linear(F.silu(x1) * x2, p[pre + "mlp.w3.weight"], p[pre + "mlp.w3.bias"])
        else:
            y = F.gelu(F.linear(y, p[pre + "mlp.fc1.weight"], p[pre + "mlp.fc1.bias"]))
            y = F.linear(y, p[pre + "mlp.fc2.weight"], p[pre + "mlp.fc2.bias"])
        return y * p[pre + "ls2.gamma"] if has_ls else y

    def residual(z: Tensor, fn, pre: str, rate: float, slot: int) -> Tensor:
        if drop is None or rate == 0.0:
            used_draws.append(None)
            return z + fn(z, pre)
        b_ = z.shape[0]
        if rate > 0.1:  # batch-subset stochastic depth, block.py:118-141
            s_ = max(int(b_ * (1 - rate)), 1)
            if drop == "torch":
                br = torch.randperm(b_)[:s_]
            else:
                kind, br = drop[slot]
                assert kind == "subset" and br.numel() == s_
            used_draws.append(("subset", br))
            res = fn(z[br], pre)
            return torch.index_add(z.flatten(1), 0, br, res.flatten(1).to(z.dtype), alpha=b_ / s_).view_as(z)
        keep = 1 - rate  # per-sample DropPath, drop_path.py:16-28
        if drop == "torch":
            sc = z.new_empty((b_,)).bernoulli_(keep)
            if keep > 0.0:
                sc.div_(keep)
        else:
            kind, sc = drop[slot]
            assert kind == "persample"
        used_draws.append(("persample", sc))
        return z + fn(z, pre) * sc.view(b_, 1, 1)

    for i in range(depth):
        pre = f"blocks.{i}."
        t = residual(t, attn_branch, pre, rates[i], 2 * i)
        t = residual(t, ffn_branch, pre, rates[i], 2 * i + 1)
        if capture is not None:
            capture[f"block{i}"] = t
    if capture is not None:
        capture["drop_draws"] = used_draws
    tn = F.layer_norm(t, (D,), p["norm.weight"], p["norm.bias"], 1e-6)
    nreg = p["register_tokens"].shape[1] if "register_tokens" in p else 0
    return {"cls": tn[:, 0], "patch": tn[:, 1 + nreg:], "prenorm": t}


BN_BUFFER_SUFFIXES = ("running_mean", "running_var", "num_batches_tracked")


def is_bn_buffer(name: str) -> bool:
    return name.endswith(BN_BUFFER_SUFFIXES)


def head_forward(p: Dict[str, Tensor], x: Tensor, buffers: Optional[Dict[str, Tensor]] = None, training: bool = True) -> Tensor:
    """DINOv2ProjectionHead.forward (LT/_methods/dinov2/dinov2_head.py:66-71).  With use_bn (dinov2_head.py:86-92) the MLP is
    Linear, BatchNorm1d, GELU, Linear, BatchNorm1d, GELU, Linear (Sequential indices 0..6): detected by the presence of mlp.1.weight.
    BatchNorm1d in train() normalises with the statistics of THIS call's rows (biased variance) and moves the running estimates
    (momentum 0.1, unbiased variance) in `buffers`; in eval() it applies the running estimates."""
    if "mlp.1.weight" in p:
        assert buffers is not None
        for lin, bn in (("mlp.0", "mlp.1"), ("mlp.3", "mlp.4")):
            x = F.linear(x, p[lin + ".weight"], p[lin + ".bias"])
            rm, rv = buffers[bn + ".running_mean"], buffers[bn + ".running_var"]
            if training:
                mean, var = x.mean(0), x.var(0, unbiased=False)
                n = x.shape[0]
                with torch.no_grad():
                    rm.mul_(0.9).add_(mean.detach(), alpha=0.1)
                    rv.mul_(0.9).add_(var.detach() * (n / (n - 1)), alpha=0.1)
                    buffers[bn + ".num_batches_tracked"] += 1
            else:
                mean, var = rm, rv
            x = F.gelu((x - mean) / torch.sqrt(var + 1e-5) * p[bn + ".weight"] + p[bn + ".bias"])
        x = F.linear(x, p["mlp.6.weight"], p["mlp.6.bias"])
    else:
        x = F.gelu(F.linear(x, p["mlp.0.weight"], p["mlp.0.bias"]))
        x = F.gelu(F.linear(x, p["mlp.2.weight"], p["mlp.2.bias"]))
        x = F.linear(x, p["mlp.4.weight"], p["mlp.4.bias"])
    x = F.normalize(x, dim=-1, p=2, eps=1e-12)
    g = p["last_layer.parametrizations.weight.original0"]
    v = p["last_layer.parametrizations.weight.original1"]
    w = v * (g / torch.linalg.vector_norm(v, dim=1, keepdim=True))  # torch._weight_norm(v, g, dim=0)
    return F.linear(x, w)


# --------------------------------------------------------------------------------------
# Losses  LT/_methods/dinov2/dinov2_loss.py
# --------------------------------------------------------------------------------------


def softmax_center(logits: Tensor, center: Tensor, temp: float) -> Tensor:
    """dinov2_loss.py:76-82 / :178-186 (center already lazily updated by the caller)."""
    return F.softmax((logits - center) / temp, dim=-1)


def center_ema(center: Tensor, batch_center_sum: Tensor, n: int, world: int, momentum: float) -> Tensor:
    """dinov2_loss.py:147-160: center*m + (sum/(n*world))*(1-m)."""
    return center * momentum + (batch_center_sum / (n * world)) * (1 - momentum)


def sinkhorn_knopp(logits: Tensor, temp: float, n_total: float, n_iter: int = 3) -> Tensor:
    """dinov2_loss.py:84-115 / :188-224, single process (world=1); n_total = B (DINO) or #masked (iBOT)."""
    q = torch.exp(logits.float() / temp).t()
    k = q.shape[0]
    q = q / q.sum()
    for _ in range(n_iter):
        q = q / q.sum(dim=1, keepdim=True)
        q = q / k
        q = q / q.sum(dim=0, keepdim=True)
        q = q / n_total
    q = q * n_total
    return q.t()


def dino_ce(student_list: Sequence[Tensor], teacher_list: Sequence[Tensor], student_temp: float) -> Tensor:
    """DINOLoss.forward dinov2_loss.py:117-133."""
    total = torch.zeros((), dtype=student_list[0].dtype, device=student_list[0].device)
    for s in student_list:
        lsm = F.log_softmax(s / student_temp, dim=-1)
        for t in teacher_list:
            total = total - (t * lsm).sum(-1).mean()
    return total


def ibot_ce_masked(s: Tensor, t: Tensor, masks_weight: Tensor, n_crops: int, student_temp: float) -> Tensor:
    """IBOTPatchLoss.forward_masked dinov2_loss.py:246-268 (B = number of global crops 2*b)."""
    per_tok = (t * F.log_softmax(s / student_temp, dim=-1)).sum(-1)
    return -(per_tok * masks_weight).sum() / n_crops


# --------------------------------------------------------------------------------------
# Optimizer param groups / schedules  LT/_methods/dinov2/utils.py:155-273, dinov2.py:550-660
# --------------------------------------------------------------------------------------


def vit_layer_id(name: str, depth: int) -> int:
    if any(s in name for s in ("pos_embed", "patch_embed", "mask_token", "cls_token", "register_tokens")):
        return 0
    if "blocks." in name and "residual." not in name:
        return int(name[name.find("blocks."):].split(".")[1]) + 1
    return depth + 1


def param_hparams(name: str, is_backbone: bool, depth: int, base_lr: float, base_wd: float,
                  layerwise_decay: float = 0.9, patch_embed_lr_mult: float = 0.2) -> Dict[str, Any]:
    """Per-parameter (lr, wd, flags) as built by get_optimizer_with_decay (utils.py:191-250)."""
    rate = layerwise_decay ** (depth + 1 - vit_layer_id(name, depth)) if is_backbone else 1.0
    lr = base_lr * rate
    wd = base_wd
    if name.endswith(".bias") or "norm" in name or "gamma" in name:
        wd = 0.0
    if "patch_embed" in name:
        lr = lr * patch_embed_lr_mult
    return {"lr": lr, "weight_decay": wd, "last_layer": "last_layer" in name, "head": "head" in name}


# --------------------------------------------------------------------------------------
# The full step
# --------------------------------------------------------------------------------------

DEFAULT_ARGS: Dict[str, Any] = dict(
    ibot_separate_head=False, hidden_dim=2048, bottleneck_dim=256, output_dim=65536,
    student_freeze_last_layer_steps=1250, dino_loss_weight=1.0, ibot_loss_weight=1.0, koleo_loss_weight=0.1,
    center_method="softmax", center_momentum=0.9, momentum_start=0.992, momentum_end=1.0,
    student_temp=0.1, teacher_temp_start=0.04, teacher_temp_end=0.07, teacher_temp_warmup_steps=37500,
    mask_ratio_min=0.1, mask_ratio_max=0.5, mask_probability=0.5, min_lr=1e-6, warmup_steps=12500,
    layerwise_decay=0.9, patch_embed_lr_multiplier=0.2, reference_batch_size=1024,
    weight_decay_start=0.04, weight_decay_end=0.4, gradient_clip_val=3.0,
    lr=0.004, betas=(0.9, 0.999), eps=1e-8,
)


class OracleDINOv2:
    """Holds student/teacher parameter dicts (reference key names) and runs the reference's
    step order: training_step_impl -> backward -> WD/lr-freeze update -> clip ->
    AdamW -> LR-scheduler -> EMA (momentum evaluated at global_step+1)."""

    def __init__(self, student_backbone: Dict[str, Tensor], student_head: Dict[str, Tensor],
                 cfg: Dict[str, Any], args: Optional[Dict[str, Any]] = None, global_batch_size: int = 16,
                 total_steps: int = 100, teacher_backbone: Optional[Dict[str, Tensor]] = None,
                 teacher_head: Optional[Dict[str, Tensor]] = None, dtype: torch.dtype = torch.float32,
                 student_ibot_head: Optional[Dict[str, Tensor]] = None, teacher_ibot_head: Optional[Dict[str, Tensor]] = None) -> None:
        self.cfg = dict(cfg)
        self.args = dict(DEFAULT_ARGS)
        self.args.update(args or {})
        cast = lambda d: {k: v.detach().clone().to(dtype) for k, v in d.items() if not is_bn_buffer(k)}  # noqa: E731
        bufs = lambda d: {k: v.detach().clone().to(torch.int64 if k.endswith("tracked") else dtype)  # noqa: E731
                          for k, v in d.items() if is_bn_buffer(k)}
        self.sb = {k: v.requires_grad_(True) for k, v in cast(student_backbone).items()}
        self.sh = {k: v.requires_grad_(True) for k, v in cast(student_head).items()}
        self.tb = cast(teacher_backbone if teacher_backbone is not None else student_backbone)
        self.th = cast(teacher_head if teacher_head is not None else student_head)
        # ibot_separate_head (dinov2.py:230-234): separate iBOT heads, else the DINO head is shared
        self.separate = student_ibot_head is not None
        self.args["ibot_separate_head"] = self.separate
        self.shi = {k: v.requires_grad_(True) for k, v in cast(student_ibot_head).items()} if self.separate else self.sh
        self.thi = cast(teacher_ibot_head if teacher_ibot_head is not None else student_ibot_head) if self.separate else self.th
        # BatchNorm1d buffers of the heads (batch_norm=True): per module, never EMA-averaged (update_momentum walks parameters() only)
        self.sh_buf = bufs(student_head)
        self.th_buf = bufs(teacher_head if teacher_head is not None else student_head)
        self.shi_buf = bufs(student_ibot_head) if self.separate else self.sh_buf
        self.thi_buf = bufs(teacher_ibot_head if teacher_ibot_head is not None else student_ibot_head) if self.separate else self.th_buf
        # freeze_eval_module(teacher_head) at construction (dinov2.py:241): the teacher heads' BatchNorm layers apply their running
        # estimates unless the caller puts the module back into train() (args["teacher_head_training"])
        self.args.setdefault("teacher_head_training", False)
        K = self.sh["last_layer.parametrizations.weight.original1"].shape[0]
        self.dino_center = torch.zeros(1, K, dtype=dtype)
        self.ibot_center = torch.zeros(1, 1, K, dtype=dtype)
        self._pending: Dict[str, Tuple[Tensor, int]] = {}
        self.global_step = 0
        self.total_steps = total_steps
        self.global_batch_size = global_batch_size
        a = self.args
        self.base_lr = a["lr"] * math.sqrt(global_batch_size / a["reference_batch_size"])
        groups = []
        for name, t in self.sb.items():
            hp = param_hparams(name, True, cfg["depth"], self.base_lr, a["weight_decay_start"],
                               a["layerwise_decay"], a["patch_embed_lr_multiplier"])
            groups.append({"name": name, "params": [t], "lr": hp["lr"], "weight_decay": hp["weight_decay"],
                           "last_layer": False})
        for name, t in self.sh.items():
            full = "dino_head." + name  # named_parameters() of DINOv2Head (shared head registered once)
            hp = param_hparams(full, False, cfg["depth"], self.base_lr, a["weight_decay_start"])
            groups.append({"name": full, "params": [t], "lr": hp["lr"], "weight_decay": hp["weight_decay"],
                           "last_layer": hp["last_layer"]})
        if self.separate:
            for name, t in self.shi.items():
                full = "ibot_head." + name
                hp = param_hparams(full, False, cfg["depth"], self.base_lr, a["weight_decay_start"])
                groups.append({"name": full, "params": [t], "lr": hp["lr"], "weight_decay": hp["weight_decay"],
                               "last_layer": hp["last_layer"]})
        self.opt = torch.optim.AdamW(groups, lr=self.base_lr, betas=a["betas"], eps=a["eps"])
        for g in self.opt.param_groups:
            g["initial_lr"] = g["lr"]
            g["wd_on"] = g["weight_decay"] != 0.0
        self.warmup = min(total_steps - 1, a["warmup_steps"])

    # ---- one forward (loss + logs), mirrors LT/_methods/dinov2/dinov2.py:259-397
    def forward_loss(self, views: List[Tensor], masks: Optional[Dict[str, Tensor]] = None,
                     capture: Optional[Dict[str, Any]] = None, drop_global: Any = "torch", drop_local: Any = "torch"
                     ) -> Tuple[Tensor, Dict[str, Tensor]]:
        a, cfg = self.args, self.cfg
        step = self.global_step
        t_temp = linear_warmup_schedule(step, a["teacher_temp_warmup_steps"], a["teacher_temp_start"], a["teacher_temp_end"])
        n_local = len(views) - 2
        terms = 2 + max(n_local * 2, 1)
        gv = torch.cat(views[:2])
        n_crops = gv.shape[0]
        b = n_crops // 2
        gh, gw = gv.shape[2] // cfg["patch_size"], gv.shape[3] // cfg["patch_size"]
        if masks is None:
            sampler = BlockMaskSampler((gh, gw), max_num_patches=int(0.5 * gh * gw))
            masks = make_collated_masks(a["mask_ratio_min"], a["mask_ratio_max"], int(n_crops * a["mask_probability"]),
                                        n_crops, sampler)
        cm, idx, mw = masks["collated_masks"], masks["mask_indices_list"], masks["masks_weight"].to(gv.dtype)
        M = idx.shape[0]
        # teacher
        with torch.no_grad():
            tt = vit_forward(self.tb, gv, cfg)
            t_cls = torch.cat([tt["cls"][b:], tt["cls"][:b]])
            tht = bool(a["teacher_head_training"])
            t_cls_logits = head_forward(self.th, t_cls, self.th_buf, tht)
            t_patch_logits = head_forward(self.thi, tt["patch"].flatten(0, 1)[idx], self.thi_buf, tht)
            if a["center_method"] == "softmax":
                self._apply_center_updates()
                t_cls_p = softmax_center(t_cls_logits, self.dino_center, t_temp).view(2, b, -1)
                self._pending["dino"] = (t_cls_logits.sum(0, keepdim=True), t_cls_logits.shape[0])
                tp = t_patch_logits.unsqueeze(0)
                t_patch_p = softmax_center(tp, self.ibot_center, t_temp).squeeze(0)
                self._pending["ibot"] = (tp.mean(1).sum(0, keepdim=True), 1)
            elif a["center_method"] == "sinkhorn_knopp":
                t_cls_p = sinkhorn_knopp(t_cls_logits, t_temp, float(t_cls_logits.shape[0])).view(2, b, -1)
                t_patch_p = sinkhorn_knopp(t_patch_logits, t_temp, float(M))
            else:
                raise ValueError(f"Unknown centering method: {a['center_method']}")
        # student global
        cap_g: Dict[str, Any] = {}
        cap_l: Dict[str, Any] = {}
        sg = vit_forward(self.sb, gv, cfg, masks=cm, drop=drop_global, capture=cap_g)
        s_cls_logits = head_forward(self.sh, sg["cls"], self.sh_buf)
        s_patch_logits = head_forward(self.shi, sg["patch"].flatten(0, 1)[idx], self.shi_buf)
        dino_global = dino_ce([s_cls_logits], [t_cls_p.flatten(0, 1)], a["student_temp"]) * 2 / terms
        dino_local = torch.zeros_like(dino_global)
        s_loc_logits = None
        if n_local > 0:
            lv = torch.cat(views[2:])
            sl = vit_forward(self.sb, lv, cfg, drop=drop_local, capture=cap_l)
            s_loc_logits = head_forward(self.sh, sl["cls"], self.sh_buf)
            dino_local = dino_ce(s_loc_logits.chunk(n_local), list(t_cls_p), a["student_temp"]) / terms
        ibot = ibot_ce_masked(s_patch_logits, t_patch_p, mw, n_crops, a["student_temp"])
        koleo = sum(koleo_loss(c) for c in sg["cls"].chunk(2))
        loss = (a["dino_loss_weight"] * (dino_global + dino_local) + a["ibot_loss_weight"] * ibot
                + a["koleo_loss_weight"] * koleo)
        if capture is not None:
            capture.update(dict(t_cls_logits=t_cls_logits, t_patch_logits=t_patch_logits, t_cls_p=t_cls_p,
                                t_patch_p=t_patch_p, s_cls_logits=s_cls_logits, s_patch_logits=s_patch_logits,
                                s_loc_logits=s_loc_logits, s_cls=sg["cls"], masks=masks, teacher_temp=t_temp,
                                drop_global=cap_g.get("drop_draws"), drop_local=cap_l.get("drop_draws")))
        logs = {"dino_global_loss": dino_global.detach(), "dino_local_loss": dino_local.detach(),
                "ibot_loss": ibot.detach(), "koleo_loss": koleo.detach()}
        return loss, logs

    def _apply_center_updates(self) -> None:
        m = self.args["center_momentum"]
        if "dino" in self._pending:
            s, n = self._pending.pop("dino")
            self.dino_center = center_ema(self.dino_center, s, n, 1, m)
        if "ibot" in self._pending:
            s, n = self._pending.pop("ibot")
            self.ibot_center = center_ema(self.ibot_center, s, n, 1, m)

    # ---- hooks in Lightning order
    def optimizer_step(self) -> Dict[str, float]:
        a, k = self.args, self.global_step
        wd = cosine_schedule(k, self.total_steps, a["weight_decay_start"], a["weight_decay_end"])
        factor = cosine_warmup_factor(k, self.warmup, self.total_steps, a["min_lr"] / self.base_lr)
        for g in self.opt.param_groups:
            g["lr"] = g["initial_lr"] * factor
            if g["wd_on"]:
                g["weight_decay"] = wd
            if g["last_layer"] and k < a["student_freeze_last_layer_steps"]:
                g["lr"] = 0.0
        params = [p for g in self.opt.param_groups for p in g["params"]]
        gnorm = torch.nn.utils.clip_grad_norm_(params, a["gradient_clip_val"])
        self.opt.step()
        self.opt.zero_grad(set_to_none=True)
        self.global_step += 1
        mom = cosine_schedule(self.global_step, self.total_steps, a["momentum_start"], a["momentum_end"])
        with torch.no_grad():
            pairs = [(self.tb, self.sb), (self.th, self.sh)] + ([(self.thi, self.shi)] if self.separate else [])
            for d_t, d_s in pairs:
                for name in d_t:
                    d_t[name].mul_(mom).add_(d_s[name].detach(), alpha=1.0 - mom)
        return {"grad_norm": float(gnorm), "weight_decay": wd, "lr_factor": factor, "momentum": mom}

    def train_step(self, views: List[Tensor], masks: Optional[Dict[str, Tensor]] = None, **kw: Any) -> Dict[str, float]:
        loss, logs = self.forward_loss(views, masks, **kw)
        loss.backward()
        info = self.optimizer_step()
        out = {k: float(v) for k, v in logs.items()}
        out["loss"] = float(loss.detach())
        out.update(info)
        return out


# --------------------------------------------------------------------------------------
# Model construction with the reference's initialisers (vision_transformer.py:224-250,
# dinov2_head.py:32-65) -- used for synthetic random-init weights on the GPU box.
# --------------------------------------------------------------------------------------

VIT_CONFIGS: Dict[str, Dict[str, Any]] = {
    "vit_test": dict(embed_dim=8, depth=3, num_heads=2, mlp_ratio=1.0),
    "vit_tiny": dict(embed_dim=192, depth=12, num_heads=3, mlp_ratio=4.0),
    "vit_small": dict(embed_dim=384, depth=12, num_heads=6, mlp_ratio=4.0),
    "vit_base": dict(embed_dim=768, depth=12, num_heads=12, mlp_ratio=4.0),
    "vit_large": dict(embed_dim=1024, depth=24, num_heads=16, mlp_ratio=4.0),
}


def init_vit_params(arch: str, patch_size: int = 16, img_size: int = 224, in_chans: int = 3,
                    init_values: float = 1e-5, generator: Optional[torch.Generator] = None, num_register_tokens: int = 0,
                    ffn_layer: str = "mlp") -> Tuple[Dict[str, Tensor], Dict[str, Any]]:
    c = dict(VIT_CONFIGS[arch])
    D, depth = c["embed_dim"], c["depth"]
    hid = int(D * c["mlp_ratio"])
    n_p = (img_size // patch_size) ** 2
    g = generator

    def tn(*shape: int) -> Tensor:
        return torch.nn.init.trunc_normal_(torch.empty(*shape), std=0.02, generator=g)

    p: Dict[str, Tensor] = {}
    p["cls_token"] = torch.empty(1, 1, D).normal_(std=1e-6, generator=g)
    p["pos_embed"] = tn(1, n_p + 1, D)
    if num_register_tokens:
        p["register_tokens"] = torch.empty(1, num_register_tokens, D).normal_(std=1e-6, generator=g)
    p["mask_token"] = torch.zeros(1, D)
    fan_in = in_chans * patch_size * patch_size
    bound = 1 / math.sqrt(fan_in)
    p["patch_embed.proj.weight"] = torch.empty(D, in_chans, patch_size, patch_size).uniform_(-bound, bound, generator=g)
    p["patch_embed.proj.bias"] = torch.empty(D).uniform_(-bound, bound, generator=g)
    for i in range(depth):
        pre = f"blocks.{i}."
        p[pre + "norm1.weight"], p[pre + "norm1.bias"] = torch.ones(D), torch.zeros(D)
        p[pre + "attn.qkv.weight"], p[pre + "attn.qkv.bias"] = tn(3 * D, D), torch.zeros(3 * D)
        p[pre + "attn.proj.weight"], p[pre + "attn.proj.bias"] = tn(D, D), torch.zeros(D)
        p[pre + "ls1.gamma"] = torch.full((D,), init_values)
        p[pre + "norm2.weight"], p[pre + "norm2.bias"] = torch.ones(D), torch.zeros(D)
        if ffn_layer in ("swiglu", "swiglufused"):
            hs = (int(hid * 2 / 3) + 7) // 8 * 8   # swiglu_ffn.py:61-63
            p[pre + "mlp.w12.weight"], p[pre + "mlp.w12.bias"] = tn(2 * hs, D), torch.zeros(2 * hs)
            p[pre + "mlp.w3.weight"], p[pre + "mlp.w3.bias"] = tn(D, hs), torch.zeros(D)
        else:
            p[pre + "mlp.fc1.weight"], p[pre + "mlp.fc1.bias"] = tn(hid, D), torch.zeros(hid)
            p[pre + "mlp.fc2.weight"], p[pre + "mlp.fc2.bias"] = tn(D, hid), torch.zeros(D)
        p[pre + "ls2.gamma"] = torch.full((D,), init_values)
    p["norm.weight"], p["norm.bias"] = torch.ones(D), torch.zeros(D)
    cfg = dict(patch_size=patch_size, num_heads=c["num_heads"], depth=depth, embed_dim=D, hidden=hid,
               img_size=img_size, in_chans=in_chans, interpolate_offset=0.1, interpolate_antialias=False)
    return p, cfg


def init_head_params(in_dim: int, hidden: int = 2048, bottleneck: int = 256, out_dim: int = 65536,
                     generator: Optional[torch.Generator] = None) -> Dict[str, Tensor]:
    g = generator

    def tn(*shape: int) -> Tensor:
        return torch.nn.init.trunc_normal_(torch.empty(*shape), std=0.02, generator=g)

    p: Dict[str, Tensor] = {}
    p["mlp.0.weight"], p["mlp.0.bias"] = tn(hidden, in_dim), torch.zeros(hidden)
    p["mlp.2.weight"], p["mlp.2.bias"] = tn(hidden, hidden), torch.zeros(hidden)
    p["mlp.4.weight"], p["mlp.4.bias"] = tn(bottleneck, hidden), torch.zeros(bottleneck)
    p["last_layer.parametrizations.weight.original0"] = torch.ones(out_dim, 1)
    bound = 1 / math.sqrt(bottleneck)  # nn.Linear default (kaiming_uniform a=sqrt(5)) == U(-1/sqrt(fan_in), +)
    p["last_layer.parametrizations.weight.original1"] = torch.empty(out_dim, bottleneck).uniform_(-bound, bound, generator=g)
    return p
