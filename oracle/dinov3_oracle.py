"""TEST INFRASTRUCTURE ONLY.  CPU restatement (plain torch fp32) of the DINOv3 ViT forward pass in eval mode, as the
distillation teacher runs it (reference LT/_models/dinov3/dinov3_src/models/vision_transformer.py:224-311,
layers/block.py:101-125, layers/attention.py:23-34,79-133, layers/rope_position_encoding.py:62-127).
Pinned against tests/golden/dinov3_vit_fwd.pt, which oracle/make_golden.py writes by running the reference's own
DinoVisionTransformer on CPU."""
from __future__ import annotations

import math
from typing import Any, Dict

import torch
import torch.nn.functional as F
from torch import Tensor


def rope_sincos(H: int, W: int, head_dim: int, base: float = 100.0, rescale: Tensor | None = None):
    """`rescale`: the training-mode log-uniform factor of rope_position_encoding.py:104-109 (a [1] tensor) or None (eval)."""
    periods = base ** (2 * torch.arange(head_dim // 4, dtype=torch.float32) / (head_dim // 2))
    ch = torch.arange(0.5, H, dtype=torch.float32) / H        # normalize_coords = "separate"
    cw = torch.arange(0.5, W, dtype=torch.float32) / W
    coords = torch.stack(torch.meshgrid(ch, cw, indexing="ij"), dim=-1).flatten(0, 1)
    coords = 2.0 * coords - 1.0
    if rescale is not None:
        coords = coords * rescale
    angles = (2 * math.pi * coords[:, :, None] / periods[None, None, :]).flatten(1, 2)
    angles = torch.cat((angles, angles), dim=-1)
    return torch.sin(angles), torch.cos(angles)


def rope_apply(x: Tensor, sin: Tensor, cos: Tensor) -> Tensor:
    x1, x2 = x.chunk(2, dim=-1)
    return x * cos + torch.cat([-x2, x1], dim=-1) * sin


def dinov3_vit_forward(p: Dict[str, Tensor], x: Tensor, cfg: Dict[str, Any], rescales: Any = None) -> Dict[str, Tensor]:
    """p: the reference state_dict (DINOv3 key names).  cfg: patch_size, num_heads, depth, rope_base, ln_eps.
    `rescales`: per-block RoPE rescale factors of a training-mode forward (list of [1] tensors) or None (eval mode)."""
    ps, heads, depth, eps = cfg["patch_size"], cfg["num_heads"], cfg["depth"], cfg.get("ln_eps", 1e-5)
    B = x.shape[0]
    t = F.conv2d(x, p["patch_embed.proj.weight"], p["patch_embed.proj.bias"], stride=ps)
    gh, gw = t.shape[2], t.shape[3]
    t = t.flatten(2).transpose(1, 2)
    n_st = p["storage_tokens"].shape[1] if "storage_tokens" in p else 0
    toks = [p["cls_token"].expand(B, -1, -1)]
    if n_st:
        toks.append(p["storage_tokens"].expand(B, -1, -1))
    t = torch.cat(toks + [t], dim=1)
    D = t.shape[-1]
    dh = D // heads
    prefix = 1 + n_st
    for i in range(depth):
        pre = f"blocks.{i}."
        sin, cos = rope_sincos(gh, gw, dh, cfg.get("rope_base", 100.0), None if rescales is None else rescales[i])
        sin, cos = sin.to(x.device), cos.to(x.device)
        y = F.layer_norm(t, (D,), p[pre + "norm1.weight"], p[pre + "norm1.bias"], eps)
        bias = p[pre + "attn.qkv.bias"]
        if pre + "attn.qkv.bias_mask" in p:
            bias = bias * p[pre + "attn.qkv.bias_mask"].to(bias.dtype)
        qkv = F.linear(y, p[pre + "attn.qkv.weight"], bias).reshape(B, -1, 3, heads, dh)
        q, k, v = [u.transpose(1, 2) for u in torch.unbind(qkv, 2)]
        q = torch.cat((q[:, :, :prefix], rope_apply(q[:, :, prefix:], sin, cos)), dim=-2)
        k = torch.cat((k[:, :, :prefix], rope_apply(k[:, :, prefix:], sin, cos)), dim=-2)
        a = ((q * dh ** -0.5) @ k.transpose(-2, -1)).softmax(dim=-1)
        y = (a @ v).transpose(1, 2).reshape(B, -1, D)
        y = F.linear(y, p[pre + "attn.proj.weight"], p[pre + "attn.proj.bias"])
        t = t + (y * p[pre + "ls1.gamma"] if pre + "ls1.gamma" in p else y)
        y = F.layer_norm(t, (D,), p[pre + "norm2.weight"], p[pre + "norm2.bias"], eps)
        y = F.linear(F.gelu(F.linear(y, p[pre + "mlp.fc1.weight"], p[pre + "mlp.fc1.bias"])), p[pre + "mlp.fc2.weight"], p[pre + "mlp.fc2.bias"])
        t = t + (y * p[pre + "ls2.gamma"] if pre + "ls2.gamma" in p else y)
    xn = F.layer_norm(t, (D,), p["norm.weight"], p["norm.bias"], eps)
    return {"x_norm_clstoken": xn[:, 0], "x_storage_tokens": xn[:, 1:prefix], "x_norm_patchtokens": xn[:, prefix:], "x_prenorm": t}
