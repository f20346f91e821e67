"""`ModelWrapper` surface of the reference for the HIP ViT engine.

Mirrors LT/_models/dinov2_vit/dinov2_vit.py:40-151 (`DINOv2ViTModelWrapper`: feature_dim, patch_size, forward_features,
forward_pool, get_model, make_teacher) and the protocol of LT/_models/model_wrapper.py:50-142.  Inference-style forward
(no activations kept); the training step (`dinov2.DINOv2`, `distillationv3.DistillationV3`) drives the engine directly.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
from torch import Tensor

from .params import FlatParams
from .vit import ViTConfig, ViTEngine, Workspace, init_vit_state, vit_param_shapes


class DINOv2ViTModelWrapper:
    def __init__(self, cfg: ViTConfig, state: Optional[Dict[str, Tensor]] = None, device: str | torch.device = "cuda",
                 params: Optional[FlatParams] = None, prefix: str = "") -> None:
        self.cfg = cfg
        if params is None:
            sd = state if state is not None else init_vit_state(cfg, torch.Generator().manual_seed(0))
            params = FlatParams([(n, sd[n]) for n, _ in vit_param_shapes(cfg)], device, False)
        self.params, self.prefix = params, prefix
        self.engine = ViTEngine(cfg, params, prefix)
        self.ws = Workspace(params.device)
        # attributes the reference reads off `get_model()` (utils.py:155-247, dinov2.py:200-203)
        self.patch_size_, self.embed_dim, self.n_blocks, self.chunked_blocks = cfg.patch_size, cfg.embed_dim, cfg.depth, False

    def feature_dim(self) -> int:
        return self.cfg.embed_dim

    def patch_size(self) -> int:
        return self.cfg.patch_size

    def forward_features(self, x: Tensor, masks: Optional[Tensor] = None) -> Dict[str, Tensor]:
        """x [B,3,H,W]; masks bool [B, n_patches] or None -> {"features": [B,D,h,w], "cls_token": [B,D]} (dinov2_vit.py:67-97).
        h, w are the patch-grid sizes the inner model uses (ceil(H/p): images that are not a multiple of the patch size are
        pad-resized like `PatchEmbed`, where the reference wrapper's `H // p` reshape fails)."""
        dev = self.params.device
        m8 = masks.to(dev).to(torch.uint8).contiguous() if masks is not None else None
        ctx = self.engine.forward(self.ws, "w", x.to(dev, torch.float32).contiguous(), m8, save=False)
        B, D, R = ctx["B"], self.cfg.embed_dim, self.cfg.num_register_tokens
        xn = ctx["xn"]
        feats = xn[:, 1 + R:].permute(0, 2, 1).reshape(B, D, ctx["gh"], ctx["gw"])
        return {"features": feats, "cls_token": xn[:, 0]}

    def forward_pool(self, x: Dict[str, Tensor]) -> Dict[str, Tensor]:
        return {"pooled_features": x["cls_token"][..., None, None]}

    def get_model(self) -> "DINOv2ViTModelWrapper":
        return self

    def make_teacher(self) -> None:
        """The reference strips drop-path from the teacher's blocks (dinov2_vit.py:108-113); the engine only applies
        stochastic depth when a drop plan is passed, which the teacher path never does."""
        return None

    def state_dict(self) -> Dict[str, Tensor]:
        n0 = len(self.prefix)
        return {n[n0:]: self.params.p[n].detach().clone() for n in self.params.names if n.startswith(self.prefix)}
