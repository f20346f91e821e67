"""TEST INFRASTRUCTURE ONLY.  Plain-torch restatement of the view-augmentation ops of the reference's DINO transform
(LT/_transforms/view_transform.py:100-215) with EXPLICIT parameters, as the checker of csrc/augment.hip.

The reference delegates the pixel work to albumentations / OpenCV (third-party, `albumentations>=1.3` / `opencv-python` in
pyproject.toml; neither vendored under /root/reference nor installed in this image): **parity unpinned** against those libraries.
Restated from their public documentation:
  * RandomResizedCrop(interpolation=cv2.INTER_AREA): crop, then resample "using pixel area relation" -- every output pixel is the mean of
    the source over its footprint (exact box filter for down-scaling, which is what INTER_AREA is used for here);
  * ColorJitter: torchvision semantics (albumentations' ColorJitter documents itself as following torchvision): brightness = factor * x,
    contrast = blend with the image-wide mean luminance, saturation = blend with the per-pixel luminance, hue = shift of H in HSV; every
    op clamps to [0, 1]; the four ops run in a random order;
  * ToGray: 0.299 R + 0.587 G + 0.114 B on all channels;  GaussianBlur: separable Gaussian, kernel radius ceil(3 sigma), reflect-101
    border (cv2's default);  Solarize(threshold t): x >= t -> 1 - x;  Normalize: (x - mean) / std."""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F
from torch import Tensor


def crop_resize_area(img_u8: Tensor, x0: float, y0: float, cw: float, ch: float, S: int, flip: bool) -> Tensor:
    """uint8 [H, W, 3] -> f32 [3, S, S] in [0, 1]."""
    H, W, _ = img_u8.shape
    img = img_u8.permute(2, 0, 1).double() / 255.0

    def weights(n_src: int, start: float, extent: float) -> Tensor:
        w = torch.zeros(S, n_src, dtype=torch.float64)
        s = extent / S
        for o in range(S):
            a, b = start + o * s, start + (o + 1) * s
            for i in range(max(0, math.floor(a)), min(n_src - 1, math.ceil(b) - 1) + 1):
                ov = min(b, i + 1) - max(a, i)
                if ov > 0:
                    w[o, i] = ov
        return w / w.sum(1, keepdim=True).clamp(min=1e-30)

    wy, wx = weights(H, y0, ch), weights(W, x0, cw)
    out = torch.einsum("oh,chw,pw->cop", wy, img, wx)
    if flip:
        out = out.flip(-1)
    return out.float()


def _lum(x: Tensor) -> Tensor:
    return 0.299 * x[0] + 0.587 * x[1] + 0.114 * x[2]


def _hue(x: Tensor, dh: float) -> Tensor:
    r, g, b = x[0], x[1], x[2]
    mx, mn = x.max(0).values, x.min(0).values
    c = mx - mn
    s = torch.where(mx > 0, c / mx.clamp(min=1e-30), torch.zeros_like(mx))
    cc = c.clamp(min=1e-30)
    h = torch.where(mx == r, (g - b) / cc, torch.where(mx == g, 2.0 + (b - r) / cc, 4.0 + (r - g) / cc)) / 6.0
    h = torch.where(c > 0, h, torch.zeros_like(h))
    h = (h - h.floor() + dh)
    h = h - h.floor()
    h6 = h * 6.0
    i = h6.floor().long() % 6
    f = h6 - h6.floor()
    v = mx
    p, q, t = v * (1 - s), v * (1 - f * s), v * (1 - (1 - f) * s)
    sel = lambda a0, a1, a2, a3, a4, a5: torch.stack([a0, a1, a2, a3, a4, a5]).gather(0, i.unsqueeze(0)).squeeze(0)  # noqa: E731
    return torch.stack([sel(v, q, p, p, t, v), sel(t, v, v, q, p, p), sel(p, p, t, v, v, q)])


def color_jitter(x: Tensor, order, fb: float, fc: float, fs: float, fh: float) -> Tensor:
    for op in order:
        if op == 0:
            x = (x * fb).clamp(0, 1)
        elif op == 1:
            x = (fc * x + (1 - fc) * _lum(x).mean()).clamp(0, 1)
        elif op == 2:
            x = (fs * x + (1 - fs) * _lum(x)).clamp(0, 1)
        else:
            x = _hue(x, fh)
    return x


def to_gray(x: Tensor) -> Tensor:
    return _lum(x).unsqueeze(0).expand(3, -1, -1).clone()


def gaussian_blur(x: Tensor, sigma: float) -> Tensor:
    if sigma <= 0:
        return x
    R = min(6, math.ceil(3 * sigma))
    k = torch.exp(-0.5 * (torch.arange(-R, R + 1, dtype=torch.float32) ** 2) / sigma ** 2)
    k = k / k.sum()
    xp = F.pad(x.unsqueeze(0), (R, R, R, R), mode="reflect")
    xp = F.conv2d(xp, k.view(1, 1, 1, -1).repeat(3, 1, 1, 1), groups=3)
    xp = F.conv2d(xp, k.view(1, 1, -1, 1).repeat(3, 1, 1, 1), groups=3)
    return xp.squeeze(0)


def finish(x: Tensor, sigma: float, solarize: bool, threshold: float, mean, std) -> Tensor:
    x = gaussian_blur(x, sigma)
    if solarize:
        x = torch.where(x >= threshold, 1 - x, x)
    return (x - torch.tensor(mean).view(3, 1, 1)) / torch.tensor(std).view(3, 1, 1)
