"""`ModelWrapper` surface of the reference for the HIP ViT engine.

Mirrors LT/_models/dinov2_vit/dinov2_vit.py:40-151 (`DINOv2ViTModelWrapper`: feature_dim, patch_size, forward_features,
forward_pool, get_model, make_teacher) and the protocol of LT/_models/model_wrapper.py:50-142.  Inference-style forward
(no activations kept); the training step (`dinov2.DINOv2`, `distillationv3.DistillationV3`) drives the engine directly.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

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
        self._model = _InnerModelView(self)
        self.activation_checkpointing, self.activation_checkpointing_every_n_blocks = False, 1

    def feature_dim(self) -> int:
        return self.cfg.embed_dim

    def patch_size(self) -> int:
        return self.cfg.patch_size

    def set_activation_checkpointing(self, enabled: bool, every_n_blocks: int = 1) -> None:
        """dinov2_vit.py:55-59.  Recorded here; the training method reads it (`DINOv2.activation_checkpointing`): this wrapper's
        own forward keeps no activations."""
        self.activation_checkpointing, self.activation_checkpointing_every_n_blocks = bool(enabled), int(every_n_blocks)

    def _intermediate(self, x: Tensor, layers: Sequence[int]) -> List[Dict[str, Tensor]]:
        """get_intermediate_layers(x, n=layers, reshape=True, return_class_token=True) (vision_transformer.py:454-480): the
        outputs of the listed blocks through the final norm, patch tokens as [B, D, h, w] + the class token."""
        dev = self.params.device
        depth = self.cfg.depth
        layers = [int(i) % depth for i in layers]
        ctx = self.engine.forward(self.ws, "w", x.to(dev, torch.float32).contiguous(), None, save=False, capture_layers=layers)
        B, D, R = ctx["B"], self.cfg.embed_dim, self.cfg.num_register_tokens
        out = []
        for i in layers:
            t = ctx["captured"][i]
            out.append({"features": t[:, 1 + R:].permute(0, 2, 1).reshape(B, D, ctx["gh"], ctx["gw"]).contiguous(), "cls_token": t[:, 0].contiguous()})
        return out

    def forward_features(self, x: Tensor, masks: Optional[Tensor] = None, n_blocks: int = 1) -> Dict[str, Tensor]:
        """x [B,3,H,W]; masks bool [B, n_patches] or None -> {"features": [B,D,h,w], "cls_token": [B,D]} (dinov2_vit.py:67-97).
        h, w are the patch-grid sizes the inner model uses (ceil(H/p): images that are not a multiple of the patch size are
        pad-resized like `PatchEmbed`, where the reference wrapper's `H // p` reshape fails).
        n_blocks > 1: channel-concatenation of the last n blocks' normed outputs, [B, n*D, h, w] / [B, n*D] (dinov2_vit.py:71-80)."""
        if n_blocks > 1:
            parts = self._intermediate(x, range(self.cfg.depth - n_blocks, self.cfg.depth))
            return {"features": torch.cat([p["features"] for p in parts], dim=1), "cls_token": torch.cat([p["cls_token"] for p in parts], dim=1)}
        dev = self.params.device
        m8 = masks.to(dev).to(torch.uint8).contiguous() if masks is not None else None
        ctx = self.engine.forward(self.ws, "w", x.to(dev, torch.float32).contiguous(), m8, save=False)
        B, D, R = ctx["B"], self.cfg.embed_dim, self.cfg.num_register_tokens
        xn = ctx["xn"]
        feats = xn[:, 1 + R:].permute(0, 2, 1).reshape(B, D, ctx["gh"], ctx["gw"])
        return {"features": feats, "cls_token": xn[:, 0]}

    def multiscale_feature_dims(self) -> List[int]:
        return [self.cfg.embed_dim] * self.cfg.depth

    def forward_multiscale_features(self, x: Tensor, layer_indices: Sequence[int]) -> List[Dict[str, Tensor]]:
        """dinov2_vit.py:118-128: one {"features", "cls_token"} per requested block index."""
        return self._intermediate(x, list(layer_indices))

    def architecture_info(self) -> Dict[str, str]:
        return {"model_type": "transformer", "norm_type": "layernorm"}

    def forward_pool(self, x: Dict[str, Tensor]) -> Dict[str, Tensor]:
        return {"pooled_features": x["cls_token"][..., None, None]}

    def get_model(self) -> "_InnerModelView":
        """The object the reference exports and reads `.patch_size` / `.embed_dim` / `.n_blocks` / `.chunked_blocks` off as
        ATTRIBUTES (dinov2.py:207, utils.py:155-247, dinov2_vit_package.py:146-162)."""
        return self._model

    def make_teacher(self) -> None:
        """The reference strips drop-path from the teacher's blocks (dinov2_vit.py:108-113); the engine only applies
        stochastic depth when a drop plan is passed, which the teacher path never does."""
        return None

    def state_dict(self) -> Dict[str, Tensor]:
        from .checkpoint import vit_key_from_flat

        n0 = len(self.prefix)
        return {vit_key_from_flat(n[n0:], self.cfg.depth, self.cfg.block_chunks): self.params.p[n].detach().clone()
                for n in self.params.names if n.startswith(self.prefix)}

    def load_state_dict(self, sd: Dict[str, Tensor], strict: bool = True) -> None:
        """Load an exported backbone (`torch.save(get_model().state_dict())`, dinov2_vit_package.py:162): chunked or plain keys."""
        from .checkpoint import vit_key_to_flat

        flat = {self.prefix + vit_key_to_flat(k): v for k, v in sd.items()}
        names = [n for n in self.params.names if n.startswith(self.prefix)]
        if strict:
            missing = [n for n in names if n not in flat]
            extra = [k for k in flat if k not in self.params.p]
            if missing or extra:
                raise KeyError(f"load_state_dict: missing {missing[:4]} unexpected {extra[:4]}")
        for n in names:
            if n in flat:
                self.params.p[n].copy_(flat[n].to(self.params.device, torch.float32))
                self.params.b[n].copy_(self.params.p[n])
        self.engine.refresh_padded_weights()


class _InnerModelView:
    """Stands where the reference's `DinoVisionTransformer` stands behind `ModelWrapper.get_model()`: attributes + state_dict."""

    def __init__(self, w: DINOv2ViTModelWrapper) -> None:
        self._w = w
        c = w.cfg
        self.patch_size, self.embed_dim, self.n_blocks, self.num_heads = c.patch_size, c.embed_dim, c.depth, c.num_heads
        self.chunked_blocks = bool(c.block_chunks)
        self.num_register_tokens = c.num_register_tokens

    def state_dict(self) -> Dict[str, Tensor]:
        return self._w.state_dict()

    def load_state_dict(self, sd: Dict[str, Tensor], strict: bool = True) -> None:
        self._w.load_state_dict(sd, strict=strict)

    def named_parameters(self):
        n0 = len(self._w.prefix)
        return [(n[n0:], self._w.params.p[n]) for n in self._w.params.names if n.startswith(self._w.prefix)]
