"""GPU multi-crop augmentation: host-side parameter sampling (CPU) and the HIP kernels against the plain-torch
restatement of the same op definitions (oracle/augment_oracle.py) with explicit parameters (-m gpu)."""
import math
import os

import numpy as np
import pytest
import torch

import lightly_train_amd  # noqa: F401
from lightly_train_amd import augment as A
from oracle import augment_oracle as AO

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_view_specs_are_the_reference_defaults():
    """DINOTransformArgs / DINOv2ViTTransformArgs (dino_transform.py:36-112, dinov2_transform.py): colour-jitter strength 0.5 x
    (0.8, 0.8, 0.4, 0.2) at p 0.8, gray p 0.2, blur p 1.0 / 0.1 / 0.5 with sigma (0.1, 2), solarize p 0.2 on global view 1."""
    specs = A.dinov2_view_specs()
    assert len(specs) == 10 and [s.size for s in specs] == [224, 224] + [98] * 8
    assert specs[0].scale == specs[1].scale == (0.32, 1.0) and specs[2].scale == (0.05, 0.32)
    assert (specs[0].blur_prob, specs[1].blur_prob, specs[2].blur_prob) == (1.0, 0.1, 0.5)
    assert (specs[0].solarize_prob, specs[1].solarize_prob, specs[2].solarize_prob) == (0.0, 0.2, 0.0)
    s = specs[0]
    assert (s.jitter_prob, s.brightness, s.contrast, s.saturation, s.hue, s.gray_prob, s.hflip_prob) == (0.8, 0.4, 0.4, 0.2, 0.1, 0.2, 0.5)


def test_crop_box_sampler_properties():
    rng = np.random.default_rng(0)
    n = 4000
    H = rng.integers(300, 500, n).astype(np.int32)      # ImageNet-like aspect ratios (the rejection step depends on them)
    W = rng.integers(300, 500, n).astype(np.int32)
    for scale in ((0.32, 1.0), (0.05, 0.32)):
        b = A.sample_crop_boxes(H, W, scale, (3 / 4, 4 / 3), rng)
        x0, y0, w, h = b.T
        assert (w >= 1).all() and (h >= 1).all() and (x0 >= 0).all() and (y0 >= 0).all()
        assert (x0 + w <= W).all() and (y0 + h <= H).all()
        frac = w * h / (H * W)
        ok = (frac > scale[0] * 0.9) & (frac < scale[1] * 1.1)
        assert ok.mean() > 0.97                                    # the rest are central-crop fallbacks of extreme aspect ratios
        r = w / h
        assert ((r > 0.7) & (r < 1.4)).mean() > 0.97
        # area fraction ~ U(scale), thinned at the top by the rejection of boxes that do not fit (torchvision's algorithm)
        assert scale[0] + 0.3 * (scale[1] - scale[0]) < frac.mean() < (scale[0] + scale[1]) / 2 + 0.02
    crop, col, fin = A.sample_view_params(A.dinov2_view_specs()[1], H, W, np.zeros(n, np.int64), rng)
    assert abs(crop["flip"].mean() - 0.5) < 0.03 and abs(col["apply"].mean() - 0.8) < 0.03 and abs(col["gray"].mean() - 0.2) < 0.03
    assert abs((fin["sigma"] > 0).mean() - 0.1) < 0.02 and abs(fin["solarize"].mean() - 0.2) < 0.03
    assert fin["sigma"].max() <= 2.0 and (fin["sigma"][fin["sigma"] > 0] >= 0.1).all()
    orders = set(int(o) for o in col["order"])
    assert len(orders) == 24 and all(sorted((o >> (2 * k)) & 3 for k in range(4)) == [0, 1, 2, 3] for o in orders)
    assert (col["fb"] >= 0.6).all() and (col["fb"] <= 1.4).all() and (np.abs(col["fh"]) <= 0.1).all()


def test_oracle_colour_ops_sanity():
    g = torch.Generator().manual_seed(0)
    x = torch.rand(3, 16, 16, generator=g)
    assert torch.allclose(AO._hue(x, 0.0), x, atol=1e-6)
    red = torch.tensor([1.0, 0.0, 0.0]).view(3, 1, 1)
    assert torch.allclose(AO._hue(red, 1 / 3), torch.tensor([0.0, 1.0, 0.0]).view(3, 1, 1), atol=1e-6)
    assert torch.allclose(AO.color_jitter(x, (0, 1, 2, 3), 1.0, 1.0, 1.0, 0.0), x, atol=1e-6)
    y = AO.color_jitter(x, (1, 0, 2, 3), 1.0, 0.0, 1.0, 0.0)        # contrast 0: everything becomes the mean luminance
    assert torch.allclose(y, AO._lum(x).mean().expand_as(y), atol=1e-6)
    img = (torch.rand(40, 60, 3, generator=g) * 255).to(torch.uint8)
    full = AO.crop_resize_area(img, 0, 0, 60, 40, 20, False)          # exact 3 x 2 box average
    ref = (img.float() / 255).permute(2, 0, 1).reshape(3, 20, 2, 20, 3).mean((2, 4))
    assert torch.allclose(full, ref, atol=1e-6)
    ident = AO.crop_resize_area(img, 10, 5, 32, 32, 32, True)
    assert torch.allclose(ident, (img[5:37, 10:42].float() / 255).permute(2, 0, 1).flip(-1), atol=1e-6)


def test_augmentation_op_definitions_by_hand():
    """albumentations / OpenCV are un-vendored (parity unpinned): every op of the restated transform evaluated on paper."""
    px = lambda *rgb: torch.tensor(rgb, dtype=torch.float32).view(3, 1, 1)   # noqa: E731
    p1 = px(0.5, 0.8, 0.2)
    lum1 = 0.299 * 0.5 + 0.587 * 0.8 + 0.114 * 0.2                          # 0.6419
    # area resampling: row (0, 255, 51) / 255 = (0, 1, .2); crop x in [0.5, 2.5) to 2 pixels: (.5*0 + .5*1, .5*1 + .5*.2) = (.5, .6); flipped (.6, .5)
    img = torch.tensor([[0, 255, 51], [0, 255, 51]], dtype=torch.uint8).unsqueeze(-1).expand(2, 3, 3).contiguous()
    out = AO.crop_resize_area(img, 0.5, 0.0, 2.0, 2.0, 2, False)
    assert torch.allclose(out[0], torch.tensor([[0.5, 0.6], [0.5, 0.6]]), atol=1e-6)
    assert torch.allclose(AO.crop_resize_area(img, 0.5, 0.0, 2.0, 2.0, 2, True)[0], torch.tensor([[0.6, 0.5], [0.6, 0.5]]), atol=1e-6)
    # a 2 x 2 crop of a 2 x 2 image to one pixel is the plain mean
    chk = torch.tensor([[0, 255], [255, 0]], dtype=torch.uint8).unsqueeze(-1).expand(2, 2, 3).contiguous()
    assert torch.allclose(AO.crop_resize_area(chk, 0, 0, 2, 2, 1, False), torch.full((3, 1, 1), 0.5), atol=1e-6)
    # brightness 1.5: (.75, 1.2 -> 1, .3)
    assert torch.allclose(AO.color_jitter(p1, (0,), 1.5, 1, 1, 0), px(0.75, 1.0, 0.3), atol=1e-6)
    # contrast 0.5 on two pixels p1 and white: mean luminance (0.6419 + 1) / 2 = 0.82095; x -> .5 x + .5 * .82095
    two = torch.cat([p1, px(1.0, 1.0, 1.0)], dim=2)
    m = (lum1 + 1.0) / 2
    want = torch.cat([px(0.25 + m / 2, 0.4 + m / 2, 0.1 + m / 2), px(0.5 + m / 2, 0.5 + m / 2, 0.5 + m / 2)], dim=2)
    assert torch.allclose(AO.color_jitter(two, (1,), 1, 0.5, 1, 0), want, atol=1e-6)
    # saturation 0: the pixel's own luminance; saturation 2: 2 x - lum, clamped: (.3581, .9581, -.2419 -> 0)
    assert torch.allclose(AO.color_jitter(p1, (2,), 1, 1, 0.0, 0), px(lum1, lum1, lum1), atol=1e-6)
    assert torch.allclose(AO.color_jitter(p1, (2,), 1, 1, 2.0, 0), px(1.0 - lum1, 1.6 - lum1, 0.0), atol=1e-6)
    # hue: (.5, .8, .2) has V = .8, S = .75, H = (2 + (.2 - .5) / .6) / 6 = .25; + .25 -> H = .5 (cyan sector, f = 0): (V(1-S), V, V) = (.2, .8, .8)
    assert torch.allclose(AO.color_jitter(p1, (3,), 1, 1, 1, 0.25), px(0.2, 0.8, 0.8), atol=1e-6)
    assert torch.allclose(AO.color_jitter(px(1.0, 0.0, 0.0), (3,), 1, 1, 1, 0.5), px(0.0, 1.0, 1.0), atol=1e-6)
    # the order matters: brightness 2 then saturation 0 -> lum(clamp(2 x)) = lum(1, 1, .4) = .9316; saturation 0 then brightness 2 -> clamp(2 * .6419) = 1
    assert torch.allclose(AO.color_jitter(p1, (0, 2), 2.0, 1, 0.0, 0), px(0.9316, 0.9316, 0.9316), atol=1e-6)
    assert torch.allclose(AO.color_jitter(p1, (2, 0), 2.0, 1, 0.0, 0), px(1.0, 1.0, 1.0), atol=1e-6)
    assert torch.allclose(AO.to_gray(p1), px(lum1, lum1, lum1), atol=1e-6)
    # Gaussian blur, sigma 0.5: radius ceil(1.5) = 2, taps exp(-2 d^2) = (1, e^-2, e^-8), normaliser Z = 1 + 2 (e^-2 + e^-8)
    Z = 1 + 2 * (math.exp(-2) + math.exp(-8))
    imp = torch.zeros(3, 7, 7); imp[:, 3, 3] = 1.0
    b = AO.gaussian_blur(imp, 0.5)
    assert b[0, 3, 3].item() == pytest.approx(1 / Z ** 2, rel=1e-5) and b[0, 3, 4].item() == pytest.approx(math.exp(-2) / Z ** 2, rel=1e-5)
    assert b[0, 2, 2].item() == pytest.approx(math.exp(-4) / Z ** 2, rel=1e-5) and b[0, 3, 6].item() == 0.0
    # reflect-101 border (column -1 mirrors to column +1, the edge pixel is not repeated): an impulse in the corner pixel (0, 0) reaches
    # (0, 0) through the centre taps only and (0, 1) through tap -1 only; an impulse at (0, 1) reaches (0, 0) through tap +1 AND the
    # mirrored tap -1: 2 e^-2 / Z in x (times 1 / Z in y)
    corner = torch.zeros(3, 7, 7); corner[:, 0, 0] = 1.0
    bc = AO.gaussian_blur(corner, 0.5)
    assert bc[0, 0, 0].item() == pytest.approx(1 / Z ** 2, rel=1e-5)
    assert bc[0, 0, 1].item() == pytest.approx(math.exp(-2) / Z ** 2, rel=1e-5)
    c1 = torch.zeros(3, 7, 7); c1[:, 0, 1] = 1.0
    assert AO.gaussian_blur(c1, 0.5)[0, 0, 0].item() == pytest.approx(2 * math.exp(-2) / Z ** 2, rel=1e-5)
    assert torch.allclose(AO.gaussian_blur(torch.full((3, 9, 9), 0.37), 1.3), torch.full((3, 9, 9), 0.37), atol=1e-6)
    # solarize at 0.5 (x >= t flips) and Normalize: ((.5, .3, .49) - mean) / std
    x = px(0.5, 0.7, 0.49)
    mean, std = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
    want = px((0.5 - 0.485) / 0.229, (0.3 - 0.456) / 0.224, (0.49 - 0.406) / 0.225)
    assert torch.allclose(AO.finish(x, 0.0, True, 0.5, mean, std), want, atol=1e-6)
    assert torch.allclose(AO.finish(x, 0.0, False, 0.5, mean, std), px((0.5 - 0.485) / 0.229, (0.7 - 0.456) / 0.224, (0.49 - 0.406) / 0.225), atol=1e-6)


@pytest.mark.gpu
def test_augmentation_kernels_match_oracle_with_explicit_parameters():
    g = torch.Generator().manual_seed(3)
    imgs = [(torch.rand(h, w, 3, generator=g) * 255).to(torch.uint8) for h, w in ((300, 400), (257, 199), (64, 64), (500, 333))]
    aug = A.GPUMultiCrop([A.ViewSpec(96, (0.3, 1.0)), A.ViewSpec(40, (0.05, 0.3))], seed=1)
    packed, H, W, off = A.GPUMultiCrop.pack(imgs)
    n = len(imgs)
    params = []
    for vi, sp in enumerate(aug.specs):
        crop, col, fin = A.sample_view_params(sp, H, W, off, np.random.default_rng(10 + vi))
        col["apply"] = [1, 1, 0, 1]; col["gray"] = [0, 1, 1, 0]              # every branch of the colour kernel
        fin["sigma"] = [0.0, 0.7, 2.0, 1.3]; fin["solarize"] = [0, 1, 0, 1]
        crop["flip"] = [0, 1, 1, 0]
        params.append((crop, col, fin))
    views = aug(packed, H, W, off, params=params)
    torch.cuda.synchronize()
    for vi, sp in enumerate(aug.specs):
        crop, col, fin = params[vi]
        for i in range(n):
            x = AO.crop_resize_area(imgs[i], float(crop["x0"][i]), float(crop["y0"][i]), float(crop["cw"][i]), float(crop["ch"][i]), sp.size,
                                    bool(crop["flip"][i]))
            if col["apply"][i]:
                order = [(int(col["order"][i]) >> (2 * k)) & 3 for k in range(4)]
                x = AO.color_jitter(x, order, float(col["fb"][i]), float(col["fc"][i]), float(col["fs"][i]), float(col["fh"][i]))
            if col["gray"][i]:
                x = AO.to_gray(x)
            ref = AO.finish(x, float(fin["sigma"][i]), bool(fin["solarize"][i]), float(fin["threshold"][i]), A.IMAGENET_MEAN, A.IMAGENET_STD)
            ours = views[vi][i].cpu()
            err = (ours - ref).abs()
            # hue / solarize are discontinuous: a value within rounding of a branch point may land on the other side -> allow a few pixels
            assert (err > 2e-3).float().mean().item() < 2e-3, (vi, i, err.max().item())
            assert err.median().item() < 1e-5


@pytest.mark.gpu
def test_gpu_multicrop_produces_the_dinov2_view_layout():
    g = torch.Generator().manual_seed(5)
    imgs = [(torch.rand(int(h), int(w), 3, generator=g) * 255).to(torch.uint8) for h, w in zip(torch.randint(200, 500, (16,), generator=g),
                                                                                               torch.randint(200, 500, (16,), generator=g))]
    aug = A.GPUMultiCrop(seed=0)
    packed, H, W, off = A.GPUMultiCrop.pack(imgs)
    views = aug(packed, H, W, off)
    assert len(views) == 10 and all(v.shape == (16, 3, 224, 224) for v in views[:2]) and all(v.shape == (16, 3, 98, 98) for v in views[2:])
    for v in views:
        assert torch.isfinite(v).all()
        assert -2.2 < float(v.min()) and float(v.max()) < 2.7          # normalised [0, 1] pixels
    # uniform-noise sources: a normalised view has mean ~ (0.5 - mean_c) / std_c per channel unless solarized / jittered: loose sanity
    m = views[0].mean((0, 2, 3)).cpu()
    assert (m - torch.tensor([(0.5 - a) / b for a, b in zip(A.IMAGENET_MEAN, A.IMAGENET_STD)])).abs().max() < 0.6
