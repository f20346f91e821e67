"""CPU: the ORCHESTRATION of the ViT engine (lightly_train_amd/vit.py: token assembly incl. masks / registers / resized positional
embedding / non-multiple image sizes, both stochastic-depth regimes, the fused LayerNorm backward that emits the next branch's gradient,
LayerScale gradients recovered from the weight gradients, the last block evaluated on the read rows only, SwiGLU) in exact arithmetic:
plain-torch stand-ins for the HIP ops (tests/tools/ops_emu.py), fp32 buffers, against torch autograd of the pinned restatement
(oracle/dinov2_oracle.py::vit_forward, itself equal to the reference to 1e-6: tests/test_oracle_pin.py).  The bf16 GPU comparisons of
the same quantities have to allow 5e-2 per tensor; here every output and every gradient tensor has to agree to 1e-4."""
import os
import sys
import types

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))

import lightly_train_amd  # noqa: E402,F401
import ops_emu  # noqa: E402
from lightly_train_amd import ops  # noqa: E402
from lightly_train_amd.params import FlatParams  # noqa: E402
from lightly_train_amd.vit import ViTConfig, ViTEngine, Workspace, init_vit_state, vit_param_shapes  # noqa: E402
from oracle import dinov2_oracle as O  # noqa: E402


class F32Workspace(Workspace):
    def get(self, name, shape, dtype, **kw):
        return super().get(name, shape, torch.float32 if dtype == torch.bfloat16 else dtype, **kw)


class _NoStream:   # the engine orders its side stream against `torch.cuda.current_stream()`; there is none here
    def record_event(self):
        return None

    def wait_event(self, ev):
        pass

    def wait_stream(self, s):
        pass


@pytest.fixture(autouse=True)
def _no_cuda_streams(monkeypatch):
    monkeypatch.setattr(torch.cuda, "current_stream", lambda *a, **k: _NoStream())


def rel(a, b):
    return ((a.float() - b.float()).abs().max() / (b.float().abs().max() + 1e-20)).item()


def make(cfg: ViTConfig, seed: int):
    g = torch.Generator().manual_seed(seed)
    sd = init_vit_state(cfg, g)
    for k in sd:       # constants of the initialiser (zero biases, unit norms, LayerScale 1e-5, zero mask token) away from their values
        if k.endswith(".bias") or "norm" in k or "gamma" in k or k in ("mask_token", "cls_token", "register_tokens"):
            sd[k] = sd[k] + 0.3 * torch.randn(sd[k].shape, generator=g)
    names = [n for n, _ in vit_param_shapes(cfg)]
    fp = FlatParams([("backbone." + n, sd[n]) for n in names], "cpu", True)
    fp.bf16 = fp.data.clone()
    fp.b = {n: fp.bf16[fp.offsets[n]:fp.offsets[n] + fp.p[n].numel()].view(fp.shapes[n]) for n in fp.names}
    eng = ViTEngine(cfg, fp, "backbone.")
    if eng.wpe_pad is not None:
        eng.wpe_pad = eng.wpe_pad.float()
        eng.refresh_padded_weights()
    params = {n: sd[n].detach().clone().requires_grad_(True) for n in names}
    return eng, fp, params, g


CASES = {
    # name: (config, batch, H, W, masks?, drop (rate, uniform) | None, read-rows-only?)
    "plain_masks": (dict(embed_dim=32, depth=2, num_heads=2, mlp_ratio=2.0, patch_size=8, img_size=32), 3, 32, 32, True, None, False),
    "registers_resized_pos_p14": (dict(embed_dim=32, depth=2, num_heads=4, mlp_ratio=2.0, patch_size=14, img_size=56, num_register_tokens=3,
                                       interpolate_antialias=True), 2, 42, 42, True, None, False),
    "non_multiple_image": (dict(embed_dim=32, depth=1, num_heads=2, mlp_ratio=2.0, patch_size=8, img_size=32), 2, 28, 30, False, None, False),
    "swiglu": (dict(embed_dim=32, depth=2, num_heads=2, mlp_ratio=4.0, patch_size=8, img_size=32, ffn_layer="swiglufused"), 2, 32, 32, True, None, False),
    "subset_stochastic_depth": (dict(embed_dim=32, depth=3, num_heads=2, mlp_ratio=2.0, patch_size=8, img_size=32, drop_path_rate=0.4,
                                     drop_path_uniform=True), 5, 32, 32, True, (0.4, True), False),
    "persample_drop_path": (dict(embed_dim=32, depth=2, num_heads=2, mlp_ratio=2.0, patch_size=8, img_size=32, drop_path_rate=0.1,
                                 drop_path_uniform=True), 6, 32, 32, False, (0.1, True), False),
    "linspace_drop_rates": (dict(embed_dim=32, depth=4, num_heads=2, mlp_ratio=2.0, patch_size=8, img_size=32, drop_path_rate=0.3), 6, 32, 32, True,
                            (0.3, False), False),                      # blocks at rate 0, 0.1 (per-sample), 0.2 and 0.3 (batch subsets)
    "no_layerscale": (dict(embed_dim=32, depth=2, num_heads=2, mlp_ratio=2.0, patch_size=8, img_size=32, init_values=None), 2, 32, 32, True, None, False),
    "activation_checkpointing": (dict(embed_dim=32, depth=3, num_heads=2, mlp_ratio=2.0, patch_size=8, img_size=32, drop_path_rate=0.4,
                                      drop_path_uniform=True), 5, 32, 32, True, (0.4, True), "checkpoint"),
    "one_input_channel": (dict(embed_dim=32, depth=1, num_heads=2, mlp_ratio=2.0, patch_size=8, img_size=32, in_chans=1), 2, 32, 32, True, None, False),
    "four_input_channels_p14": (dict(embed_dim=32, depth=1, num_heads=2, mlp_ratio=2.0, patch_size=14, img_size=28, in_chans=4), 2, 28, 28, False, None, False),
    "last_block_on_read_rows": (dict(embed_dim=32, depth=2, num_heads=2, mlp_ratio=2.0, patch_size=8, img_size=32), 4, 32, 32, True, None, True),
    "read_rows_with_subset_depth": (dict(embed_dim=32, depth=2, num_heads=2, mlp_ratio=2.0, patch_size=8, img_size=32, drop_path_rate=0.5,
                                         drop_path_uniform=True), 4, 32, 32, True, (0.5, True), True),
}


@pytest.mark.parametrize("name", list(CASES))
def test_vit_engine_forward_backward_equals_autograd_of_the_restatement(name):
    ck, B, H, W, use_masks, drop, sparse = CASES[name]
    cfg = ViTConfig(**ck)
    with ops_emu.emulate(ops):
        eng, fp, params, g = make(cfg, seed=len(name))
        p, D, nreg = cfg.patch_size, cfg.embed_dim, cfg.num_register_tokens
        img = torch.randn(B, cfg.in_chans, H, W, generator=g)
        gh, gw = -(-H // p), -(-W // p)
        n_p, N = gh * gw, gh * gw + 1 + nreg
        masks = None
        if use_masks:
            masks = torch.rand(B, n_p, generator=g) < 0.3
            masks[-1] = False                                        # an unmasked crop among masked ones
        ocfg = dict(patch_size=p, num_heads=cfg.num_heads, depth=cfg.depth, interpolate_offset=cfg.interpolate_offset,
                    interpolate_antialias=cfg.interpolate_antialias, drop_path_rate=cfg.drop_path_rate, drop_path_uniform=cfg.drop_path_uniform)
        cap = {}
        torch.manual_seed(7)
        out = O.vit_forward(params, img, ocfg, masks=masks, capture=cap, drop="torch" if drop else None)
        plan = cap["drop_draws"] if drop else None
        if drop:
            assert any(e is not None for e in plan)
        # upstream gradient of the final-norm tokens (registers are never read by a loss)
        dxn = torch.randn(B, N, D, generator=g)
        dxn[:, 1:1 + nreg] = 0
        rows = None
        fkw = {}
        if sparse == "checkpoint":   # blocks recomputed one at a time during backward (ViTEngine.forward(checkpoint=True))
            fkw, sparse = dict(checkpoint=True), False
        if sparse:   # the DINOv2 losses read the cls rows and the masked patch rows only
            keep = torch.zeros(B, N, dtype=torch.bool)
            keep[:, 0] = True
            keep[:, 1 + nreg:] = masks
            dxn = dxn * keep.unsqueeze(-1)
            idx = keep.flatten().nonzero().flatten()
            rows = (torch.cat([idx, idx.new_zeros(5)]), int(idx.numel()))        # padded index buffer, R valid entries
        ((out["cls"] * dxn[:, 0]).sum() + (out["patch"] * dxn[:, 1 + nreg:]).sum()).backward()

        ws = F32Workspace(torch.device("cpu"))
        ctx = eng.forward(ws, "s", img, masks.to(torch.uint8) if masks is not None else None, save=True, drop_plan=plan, last_mlp_rows=rows, **fkw)
        xn = ctx["xn"].view(B, N, D)
        fp.grad.zero_()
        eng.backward(ws, ctx, dxn.reshape(B * N, D).clone())
        eng.finish_layerscale_grads()
    want = torch.cat([out["cls"].detach().unsqueeze(1), torch.zeros(B, nreg, D), out["patch"].detach()], dim=1)
    sel = torch.ones(B, N, dtype=torch.bool)
    sel[:, 1:1 + nreg] = False
    if sparse:
        sel &= keep
    assert torch.allclose(xn[sel], want[sel], atol=2e-5, rtol=1e-4), (xn[sel] - want[sel]).abs().max()
    worst = ("", 0.0)
    for n_, t in params.items():
        mine = fp.g["backbone." + n_]
        if t.grad is None:
            assert mine.abs().max().item() == 0, n_
            continue
        e = rel(mine, t.grad) if t.grad.abs().max() > 0 else mine.abs().max().item()
        worst = max(worst, (n_, e), key=lambda z: z[1])
        assert e < 1e-4, (n_, e)


def test_model_wrapper_surface_in_exact_arithmetic():
    """ModelWrapper surface (LT/_models/dinov2_vit/dinov2_vit.py:55-128): forward_features with iBOT masks, forward_pool, n_blocks > 1,
    forward_multiscale_features, the state_dict round trip of the exported backbone -- against the restatement, fp32 round-off."""
    from lightly_train_amd.model_wrapper import DINOv2ViTModelWrapper

    fx = torch.load(os.path.join(ROOT, "tests", "golden", "step_d64_softmax.pt"), weights_only=False)
    sb = fx["init"]["student_backbone"]
    cfg = ViTConfig(embed_dim=64, depth=2, num_heads=1, mlp_ratio=4.0, patch_size=16, img_size=fx["g_size"])
    ocfg = dict(patch_size=16, num_heads=1, depth=2)
    with ops_emu.emulate(ops):
        def exact(w):
            w.ws = F32Workspace(torch.device("cpu"))
            w.params.bf16 = w.params.data.clone()
            w.params.b = {n: w.params.bf16[w.params.offsets[n]:w.params.offsets[n] + w.params.p[n].numel()].view(w.params.shapes[n]) for n in w.params.names}
            return w

        w = exact(DINOv2ViTModelWrapper(cfg, state=sb, device="cpu"))
        g = torch.Generator().manual_seed(11)
        x = torch.randn(3, 3, 96, 96, generator=g)
        masks = torch.rand(3, 36, generator=g) < 0.3
        out = {k: v.clone() for k, v in w.forward_features(x, masks).items()}
        ref = O.vit_forward(sb, x, ocfg, masks=masks)
        assert out["features"].shape == (3, 64, 6, 6)
        assert torch.allclose(out["cls_token"], ref["cls"], atol=2e-5)
        assert torch.allclose(out["features"].flatten(2).transpose(1, 2), ref["patch"], atol=2e-5)
        assert torch.allclose(w.forward_pool(out)["pooled_features"].flatten(1), ref["cls"], atol=2e-5)     # the cls token (dinov2_vit.py:99-103)
        cap = {}
        O.vit_forward(sb, x, ocfg, capture=cap)
        normed = [torch.nn.functional.layer_norm(cap[f"block{i}"], (64,), sb["norm.weight"], sb["norm.bias"], 1e-6) for i in range(2)]
        ms = [{k: v.clone() for k, v in d_.items()} for d_ in w.forward_multiscale_features(x, [0, 1])]
        for i in range(2):
            assert torch.allclose(ms[i]["cls_token"], normed[i][:, 0], atol=2e-5)
            assert torch.allclose(ms[i]["features"].flatten(2).transpose(1, 2), normed[i][:, 1:], atol=2e-5)
        cat = w.forward_features(x, n_blocks=2)
        assert torch.allclose(cat["cls_token"], torch.cat([normed[0][:, 0], normed[1][:, 0]], 1), atol=2e-5)
        w2 = exact(DINOv2ViTModelWrapper(cfg, device="cpu"))
        w2.get_model().load_state_dict(w.get_model().state_dict())
        w2.params.bf16.copy_(w2.params.data)
        assert torch.equal(w2.forward_features(x, masks)["cls_token"], out["cls_token"])
