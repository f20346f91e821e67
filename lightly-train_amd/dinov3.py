"""DINOv3 ViT (frozen distillation teacher) on the same HIP engine as the DINOv2 ViT.

Reference: LT/_models/dinov3/dinov3_src/models/vision_transformer.py:75-320 (DinoVisionTransformer),
layers/attention.py:23-133 (SelfAttention + RoPE), layers/rope_position_encoding.py:62-127, dinov3_vit.py:40-90 (wrapper).
Differences from the DINOv2 ViT that matter for the forward pass: no learned position embedding -- rotary embedding on q / k
of the patch tokens in every block; `storage_tokens` instead of `register_tokens`; LayerNorm eps 1e-5 ("layernormbf16");
the K third of the qkv bias is masked to zero (`LinearKMaskedBias`).  The engine state keeps the DINOv2 key names, so a DINOv3
state dict is converted once at load time.
"""
from __future__ import annotations

from typing import Any, Dict

import torch
from torch import Tensor

from .vit import ViTConfig


def dinov3_vit_config(embed_dim: int, depth: int, num_heads: int, patch_size: int = 16, img_size: int = 224, ffn_ratio: float = 4.0,
                      n_storage_tokens: int = 4, layerscale_init: float | None = 1e-5, rope_base: float = 100.0,
                      ln_eps: float = 1e-5, ffn_layer: str = "mlp", rope_rescale: float | None = None,
                      mask_k_bias: bool = True) -> ViTConfig:
    """`dinov3_vitl16` = (1024, 24, 16) with the defaults (hub/backbones.py:467-512).  `rope_rescale` (2 in those recipes)
    only matters for a *student* (training-mode coordinate augmentation); a frozen teacher runs in eval mode."""
    return ViTConfig(embed_dim=embed_dim, depth=depth, num_heads=num_heads, mlp_ratio=ffn_ratio, patch_size=patch_size, img_size=img_size,
                     init_values=layerscale_init, num_register_tokens=n_storage_tokens, ffn_layer=ffn_layer, ln_eps=ln_eps,
                     rope_base=rope_base, rope_rescale=rope_rescale, mask_k_bias=mask_k_bias)


def export_dinov3_state(engine_state: Dict[str, Tensor], cfg: ViTConfig) -> Dict[str, Tensor]:
    """Inverse of `convert_dinov3_state` for checkpoints: register_tokens -> storage_tokens, pos_embed dropped, the bias mask
    and the RoPE periods re-created."""
    out: Dict[str, Tensor] = {}
    D, dh = cfg.embed_dim, cfg.head_dim
    for k, v in engine_state.items():
        if k == "pos_embed":
            continue
        if k == "register_tokens":
            out["storage_tokens"] = v
        else:
            out[k] = v
        if k.endswith("attn.qkv.bias") and cfg.mask_k_bias:
            m = torch.ones(3 * D)
            m[D:2 * D] = 0
            out[k + "_mask"] = m
    out["rope_embed.periods"] = float(cfg.rope_base) ** (2 * torch.arange(dh // 4, dtype=torch.float32) / (dh // 2))
    return out


def convert_dinov3_state(state: Dict[str, Tensor], cfg: ViTConfig) -> Dict[str, Tensor]:
    """Reference DINOv3 `state_dict()` -> the engine's (DINOv2-named) backbone state: storage_tokens -> register_tokens, the
    qkv bias multiplied by its mask (layers/attention.py:37-57), a zero pos_embed, `rope_embed.periods` dropped (the sin / cos
    tables are rebuilt from `cfg.rope_base`, checked here against the stored periods)."""
    out: Dict[str, Tensor] = {}
    D = cfg.embed_dim
    for k, v in state.items():
        if k == "storage_tokens":
            out["register_tokens"] = v.detach().clone().float()
        elif k == "rope_embed.periods":
            dh = cfg.head_dim
            expect = float(cfg.rope_base) ** (2 * torch.arange(dh // 4, dtype=torch.float32) / (dh // 2))
            if not torch.allclose(v.float().cpu(), expect, rtol=1e-2):
                raise ValueError("rope_embed.periods of the checkpoint do not match base ** (2 i / (head_dim / 2)) for cfg.rope_base")
        elif k.endswith("attn.qkv.bias_mask"):
            continue
        elif k.endswith("attn.qkv.bias"):
            mask = state.get(k + "_mask")
            b = v.detach().clone().float()
            if mask is not None:
                b = b * mask.to(b.dtype).nan_to_num(nan=0.0)
            out[k] = b
        elif k.startswith("local_cls_norm."):
            # untie_global_and_local_cls_norm (the SAT-493M ViT-L and the ViT-7B recipes, hub/backbones.py:476-487,615): a LayerNorm applied to
            # the class / storage tokens of the LOCAL crops of a TRAINING forward only (vision_transformer.py:286-292).  The frozen teacher
            # of the distillation methods runs in eval(): the parameters are never read there, so they are dropped on load.
            continue
        elif k.startswith(("cls_norm.", "head.")):
            raise NotImplementedError(f"untie_cls_and_patch_norms / classification heads are not supported by the engine ({k})")
        else:
            out[k] = v.detach().clone().float()
    n_p = (cfg.img_size // cfg.patch_size) ** 2
    out["pos_embed"] = torch.zeros(1, n_p + 1, D)
    return out
