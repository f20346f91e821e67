"""TEST INFRASTRUCTURE ONLY -- restatements for the DINOv31 method (LT/_methods/dinov31/dinov31.py: DINOv2 + the PaKA dense-relational loss).

**PARITY UNPINNED** for the two LightlySSL pieces the method imports and the reference tree does not contain (`lightly>=1.5.26`,
`from lightly.loss import PatchKernelAlignmentLoss, roi_resample_to_grid`, dinov31.py:55): they are restated here from what the
reference's own code fixes about them --

  * `roi_resample_to_grid(feat [B, C, H, W], boxes [B, 4] = (x0, y0, x1, y1) in feature-grid units, out_h, out_w) -> [B, out_h * out_w, C]`
    (call site dinov31.py:423-437: boxes are clamped to [0, W] x [0, H], i.e. cell j of the map spans [j, j + 1)): every output cell samples
    the map bilinearly at the centre of its bin of the box (RoIAlign with one sample per bin, `aligned=True` convention: map cell centres sit
    at j + 0.5), border-clamped;
  * `PatchKernelAlignmentLoss(max_tokens)(student_features [B, N, C], teacher_features [B, N, C], mask [B, N] True = ignore)` (call site
    dinov31.py:330-336; docstring :8-16 "Patch Kernel Alignment (PaKA / CKA) loss that aligns the relational structure of student and
    teacher dense patch tokens", "averages over the pairs that actually overlap", `paka_max_tokens`: "Per-image token subsample before the
    O(N^2) CKA kernel"): per image the linear-kernel centred kernel alignment of the two token Gram matrices,
        K = Z Z^T,  Kc = H K H (H = I - 11^T / n),  CKA = <Kc_s, Kc_t>_F / (||Kc_s||_F ||Kc_t||_F + eps),  loss = mean_valid (1 - CKA),
    over the unmasked tokens of each image (images with fewer than 2 of them are skipped; no valid image: 0), a uniform random subset of
    `max_tokens` tokens when an image has more.

What IS pinned (tests/golden/dinov31_d64.pt, oracle/make_golden.py --dinov31): everything dinov31.py itself does around them -- the view
split, the clean-teacher pass, parent-only pairing, the shared-region / flip / box arithmetic of `_align_cross_view_pair` and
`_roi_align_view` (in-tree code), the PaKA heads and their EMA, `paka_weight`, `paka_start_step` -- by running the reference's own `DINOv31`
class on these two restatements (oracle/ref_harness.py registers them as `lightly.loss.*`, like LARS and the DINO v1 pieces).
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import Tensor


def roi_resample_to_grid(feat: Tensor, boxes: Tensor, out_h: int, out_w: int) -> Tensor:
    """feat [B, C, H, W], boxes [B, 4] (x0, y0, x1, y1) in grid units -> [B, out_h * out_w, C] (row-major over the output grid)."""
    B, C, H, W = feat.shape
    x0, y0, x1, y1 = boxes[:, 0], boxes[:, 1], boxes[:, 2], boxes[:, 3]
    ox = (torch.arange(out_w, dtype=feat.dtype, device=feat.device) + 0.5) / out_w
    oy = (torch.arange(out_h, dtype=feat.dtype, device=feat.device) + 0.5) / out_h
    sx = x0[:, None] + ox[None, :] * (x1 - x0)[:, None] - 0.5      # [B, out_w] in cell-index space (centre of cell j = j)
    sy = y0[:, None] + oy[None, :] * (y1 - y0)[:, None] - 0.5      # [B, out_h]
    sx = sx.clamp(0.0, W - 1.0)
    sy = sy.clamp(0.0, H - 1.0)
    xl, yl = sx.floor(), sy.floor()
    xh, yh = (xl + 1).clamp(max=W - 1.0), (yl + 1).clamp(max=H - 1.0)
    wx, wy = sx - xl, sy - yl
    xl, xh, yl, yh = xl.long(), xh.long(), yl.long(), yh.long()
    bi = torch.arange(B, device=feat.device)[:, None, None]
    f = feat.permute(0, 2, 3, 1)                                   # [B, H, W, C]

    def at(yy: Tensor, xx: Tensor) -> Tensor:
        return f[bi, yy[:, :, None], xx[:, None, :]]               # [B, out_h, out_w, C]

    wxb, wyb = wx[:, None, :, None], wy[:, :, None, None]
    out = (at(yl, xl) * (1 - wyb) * (1 - wxb) + at(yl, xh) * (1 - wyb) * wxb + at(yh, xl) * wyb * (1 - wxb) + at(yh, xh) * wyb * wxb)
    return out.reshape(B, out_h * out_w, C)


def resample_tables(boxes: Tensor, H: int, W: int, out_h: int, out_w: int):
    """The same sampling as 4-tap tables: idx int32 [B, out_h * out_w, 4] into the flattened H x W map and weights f32 of the same shape
    (what the HIP path feeds lt_resample_tokens_batched; shared arithmetic so that kernel and restatement cannot drift)."""
    B = boxes.shape[0]
    x0, y0, x1, y1 = boxes[:, 0], boxes[:, 1], boxes[:, 2], boxes[:, 3]
    ox = (torch.arange(out_w, dtype=boxes.dtype) + 0.5) / out_w
    oy = (torch.arange(out_h, dtype=boxes.dtype) + 0.5) / out_h
    sx = (x0[:, None] + ox[None, :] * (x1 - x0)[:, None] - 0.5).clamp(0.0, W - 1.0)
    sy = (y0[:, None] + oy[None, :] * (y1 - y0)[:, None] - 0.5).clamp(0.0, H - 1.0)
    xl, yl = sx.floor(), sy.floor()
    xh, yh = (xl + 1).clamp(max=W - 1.0), (yl + 1).clamp(max=H - 1.0)
    wx, wy = (sx - xl)[:, None, :], (sy - yl)[:, :, None]
    xl, xh, yl, yh = xl.long()[:, None, :], xh.long()[:, None, :], yl.long()[:, :, None], yh.long()[:, :, None]
    idx = torch.stack([yl * W + xl, yl * W + xh, yh * W + xl, yh * W + xh], dim=-1)           # [B, out_h, out_w, 4]
    wts = torch.stack([(1 - wy) * (1 - wx), (1 - wy) * wx, wy * (1 - wx), wy * wx], dim=-1)
    return idx.reshape(B, out_h * out_w, 4).to(torch.int32), wts.reshape(B, out_h * out_w, 4).to(torch.float32)


def linear_cka(zs: Tensor, zt: Tensor, eps: float = 1e-8) -> Tensor:
    """Centred kernel alignment of the token Gram matrices of one image: zs, zt [n, C]."""
    zs = zs - zs.mean(0, keepdim=True)          # H Z: centring the kernel = centring the features over the tokens
    zt = zt - zt.mean(0, keepdim=True)
    ks, kt = zs @ zs.t(), zt @ zt.t()
    return (ks * kt).sum() / (ks.norm() * kt.norm() + eps)


class PatchKernelAlignmentLoss(torch.nn.Module):
    def __init__(self, max_tokens: int = 512, eps: float = 1e-8) -> None:
        super().__init__()
        self.max_tokens, self.eps = int(max_tokens), float(eps)

    def forward(self, student_features: Tensor, teacher_features: Tensor, mask: Optional[Tensor] = None) -> Tensor:
        B = student_features.shape[0]
        terms = []
        for b in range(B):
            keep = torch.ones(student_features.shape[1], dtype=torch.bool, device=student_features.device) if mask is None else ~mask[b]
            idx = keep.nonzero().flatten()
            if idx.numel() < 2:
                continue
            if idx.numel() > self.max_tokens:
                idx = idx[torch.randperm(idx.numel(), device=idx.device)[: self.max_tokens]]
            terms.append(1.0 - linear_cka(student_features[b, idx], teacher_features[b, idx].detach(), self.eps))
        if not terms:
            return student_features.sum() * 0.0
        return torch.stack(terms).mean()
