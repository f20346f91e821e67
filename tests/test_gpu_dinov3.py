"""-m gpu: the DINOv3 ViT forward (frozen distillation teacher: RoPE, storage tokens, K-masked qkv bias, eps 1e-5) on the HIP
engine against outputs of the reference's own model (tests/golden/dinov3_vit_fwd.pt) and the RoPE kernel against torch."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-20)).item()


def test_rope_kernel_matches_reference_formula_and_inverts():
    import lightly_train_amd  # noqa: F401
    from lightly_train_amd import ops
    from oracle import dinov3_oracle as O3

    B, H, dh, gh, gw, prefix = 2, 3, 64, 3, 5, 5
    N = prefix + gh * gw
    g = torch.Generator().manual_seed(3)
    qkv = torch.randn(B, N, 3, H, dh, generator=g).to(torch.bfloat16)
    sin, cos = O3.rope_sincos(gh, gw, dh, 100.0)
    ref = qkv.float().clone()
    for w in (0, 1):
        x = ref[:, prefix:, w].permute(0, 2, 1, 3)                    # [B, H, hw, dh]
        ref[:, prefix:, w] = O3.rope_apply(x, sin, cos).permute(0, 2, 1, 3)
    d = qkv.cuda().contiguous()
    ops.rope_apply(d, sin.cuda(), cos.cuda(), B, N, H, dh, prefix)
    assert rel(d, ref) < 6e-3
    assert torch.equal(d[:, :prefix].cpu(), qkv[:, :prefix]) and torch.equal(d[:, :, 2].cpu(), qkv[:, :, 2])   # prefix tokens and v untouched
    ops.rope_apply(d, sin.cuda(), cos.cuda(), B, N, H, dh, prefix, inverse=True)     # rotation^T o rotation = identity
    assert rel(d, qkv) < 1.2e-2


def test_dinov3_vit_forward_matches_reference_fixture():
    import lightly_train_amd  # noqa: F401
    from lightly_train_amd.dinov3 import convert_dinov3_state, dinov3_vit_config
    from lightly_train_amd.params import FlatParams
    from lightly_train_amd.vit import ViTEngine, Workspace, vit_param_shapes

    fx = torch.load(os.path.join(GOLD, "dinov3_vit_fwd.pt"), weights_only=False)
    c = fx["cfg"]
    cfg = dinov3_vit_config(c["embed_dim"], c["depth"], c["num_heads"], patch_size=c["patch_size"], img_size=c["img_size"],
                            n_storage_tokens=c["n_storage_tokens"], layerscale_init=0.5, rope_base=c["rope_base"], ln_eps=c["ln_eps"])
    sd = convert_dinov3_state(fx["state"], cfg)
    fp = FlatParams([(n, sd[n]) for n, _ in vit_param_shapes(cfg)], "cuda", False)
    eng = ViTEngine(cfg, fp, "")
    ws = Workspace(torch.device("cuda"))
    R = c["n_storage_tokens"]
    for case in fx["cases"]:
        x = torch.randn(*case["shape"], generator=torch.Generator().manual_seed(case["seed"]))
        ctx = eng.forward(ws, "t", x.cuda(), None, save=False)
        xn = ctx["xn"]
        assert rel(xn[:, 0], case["out"]["x_norm_clstoken"]) < 2e-2
        assert rel(xn[:, 1:1 + R], case["out"]["x_storage_tokens"]) < 2e-2
        assert rel(xn[:, 1 + R:], case["out"]["x_norm_patchtokens"]) < 2e-2
