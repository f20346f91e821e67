"""CPU: the DINOv31 step (lightly_train_amd/dinov31.py: DINOv2 + the PaKA dense-relational term, LT/_methods/dinov31/dinov31.py) in exact
arithmetic -- the HIP kernels replaced by the plain-torch statements of their contracts (tests/tools/ops_emu.py) -- against the fixture the
REFERENCE's own `DINOv31` class wrote (tests/golden/dinov31_d64.pt, oracle/make_dinov31_fixture.py): step 0 without PaKA
(`paka_start_step = 1`), steps 1-2 with it.  Loss terms incl. `paka_loss`, the gradient norm, the PaKA head's gradients, and after the
three optimizer steps every student / EMA-teacher tensor incl. both PaKA heads.  Also: the restated RoI sampling agrees with its table
form, and the host-side geometry equals the reference's `_align_cross_view_pair` arithmetic (through the fixture's loss values)."""
import os
import random
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import lightly_train_amd  # noqa: E402,F401
import ops_emu  # noqa: E402
from lightly_train_amd import ops  # noqa: E402
from lightly_train_amd.dinov31 import DINOv31, DINOv31Args, init_paka_head_state, roi_tables  # noqa: E402
from lightly_train_amd.vit import ViTConfig  # noqa: E402
from test_distillation_methods_cpu import _NoStream  # noqa: E402
from test_integration_cpu import exactify  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(autouse=True)
def _no_cuda_streams(monkeypatch):
    import contextlib

    monkeypatch.setattr(torch.cuda, "current_stream", lambda *a, **k: _NoStream())
    monkeypatch.setattr(torch.cuda, "set_stream", lambda s: None)
    monkeypatch.setattr(torch.cuda, "Stream", _NoStream)
    monkeypatch.setattr(torch.cuda, "stream", lambda s: contextlib.nullcontext())


def synth_views(fx, seed):
    g = torch.Generator().manual_seed(seed)
    b, G, L = fx["b"], fx["g_size"], fx["l_size"]
    v = [torch.randn(b, 3, G, G, generator=g) for _ in range(2)] + [torch.randn(b, 3, L, L, generator=g) for _ in range(fx["n_local"])]
    return v + [torch.randn(b, 3, G, G, generator=g) for _ in range(2)] + [torch.randn(b, 3, L, L, generator=g) for _ in range(fx["k_paka"])]


def build(fx, device="cpu"):
    mk, cfgd, init = fx["method_kwargs"], fx["cfg"], fx["init"]
    sb = init["student_backbone"]
    D = sb["cls_token"].shape[-1]
    vc = ViTConfig(embed_dim=D, depth=cfgd["depth"], num_heads=cfgd["num_heads"], patch_size=cfgd["patch_size"], img_size=fx["g_size"], init_values=cfgd["init_values"],
                   mlp_ratio=sb["blocks.0.mlp.fc1.weight"].shape[0] / D)
    ph = init_paka_head_state(D, torch.Generator().manual_seed(fx["paka_seed"]))
    args = DINOv31Args(**mk)
    return DINOv31(vc, args, global_batch_size=fx["b"], total_steps=fx["total_steps"], device=device, backbone_state=sb, student_head_state=init["student_head"],
                   teacher_head_state=init["teacher_head"], paka_head_state=ph)


def strided(t):
    return t if t.numel() <= 4096 else t.reshape(t.shape[0], -1)[::16, ::8]


def test_roi_tables_equal_the_restated_sampling():
    from oracle import dinov31_oracle as O31

    g = torch.Generator().manual_seed(0)
    B, C, H, W, oh, ow = 5, 7, 6, 9, 3, 4
    feat = torch.randn(B, C, H, W, generator=g)
    x0 = torch.rand(B, generator=g) * 4; y0 = torch.rand(B, generator=g) * 3
    boxes = torch.stack([x0, y0, x0 + 1 + torch.rand(B, generator=g) * 4, y0 + 1 + torch.rand(B, generator=g) * 2.5], 1)
    want = O31.roi_resample_to_grid(feat, boxes, oh, ow)
    geom = torch.zeros(B, 8)
    idx, w = roi_tables(boxes, geom, (H, W), (oh, ow))
    flat = feat.permute(0, 2, 3, 1).reshape(B, H * W, C)
    got = torch.stack([(flat[b][idx[b].long()] * w[b][:, :, None]).sum(1) for b in range(B)])
    assert torch.allclose(got, want, atol=1e-6)
    # flips: the reference flips the map first (dinov31.py:419-422), the tables mirror the indices instead
    geom[:, 6] = torch.tensor([1.0, 0, 1, 0, 1]); geom[:, 7] = torch.tensor([0.0, 1, 1, 0, 0])
    f2 = torch.where((geom[:, 6] > 0.5)[:, None, None, None], feat.flip(-1), feat)
    f2 = torch.where((geom[:, 7] > 0.5)[:, None, None, None], f2.flip(-2), f2)
    want = O31.roi_resample_to_grid(f2.contiguous(), boxes, oh, ow)
    idx, w = roi_tables(boxes, geom, (H, W), (oh, ow))
    got = torch.stack([(flat[b][idx[b].long()] * w[b][:, :, None]).sum(1) for b in range(B)])
    assert torch.allclose(got, want, atol=1e-6)


def test_dinov31_three_steps_equal_the_reference_class():
    fx = torch.load(os.path.join(GOLD, "dinov31_d64.pt"), weights_only=False)
    with ops_emu.emulate(ops):
        m = build(fx)
        exactify(m)
        assert [n for n in m.student.names if n.startswith("paka.")] == [f"paka.{l}.{p}" for l in ("0", "2", "4") for p in ("weight", "bias")]
        for si, rec in enumerate(fx["steps"]):
            views = synth_views(fx, rec["seed"])
            res = m.training_step_impl({"views": views, "geometries": rec["geometries"]}, si, masks=rec["masks"])
            logs = {k.split("/")[-1]: float(v) for k, v in res.log_dict.items()}
            assert ("paka_loss" in logs) == (si >= 1)
            for k in ("dino_global_loss", "dino_local_loss", "ibot_loss") + (("paka_loss",) if si >= 1 else ()):
                assert logs[k] == pytest.approx(rec["logs"][k], rel=5e-5, abs=5e-5), (si, k, logs[k], rec["logs"][k])
            assert float(res.loss) == pytest.approx(rec["logs"]["loss"], rel=5e-5)
            if si >= 1:
                for k, ref in rec["paka_grad"].items():
                    ours = strided(m.student.g["paka." + k])
                    if k == "4.bias":   # a bias in front of the centring over tokens has NO gradient: summation round-off on both sides
                        assert ours.abs().max().item() < 1e-7 and ref.abs().max().item() < 1e-7
                        continue
                    assert (ours.reshape(ref.shape) - ref).abs().max().item() <= 2e-3 * ref.abs().max().item() + 1e-9, (si, k)
                    assert float(m.student.g["paka." + k].norm()) == pytest.approx(rec["paka_grad_norm"][k], rel=2e-3), (si, k)
            else:
                assert float(m.student.g["paka.0.weight"].abs().max()) == 0.0
            m.optimizer_step()
            assert float(m.last_grad_norm.sqrt()) == pytest.approx(rec["logs"]["grad_norm"], rel=2e-3), si
            m.on_train_batch_end()
        sd = m.state_dict()
        last = fx["steps"][-1]
        assert [k for k in sd if "_paka_head." not in k] == list(last["state"])
        assert [k for k in sd if "_paka_head." in k] == list(last["paka_state"])
        for k, ref in last["state"].items():
            assert torch.allclose(sd[k].float(), ref.float(), atol=5e-5), (k, (sd[k].float() - ref.float()).abs().max().item())
        for k, ref in last["paka_state"].items():
            assert torch.allclose(strided(sd[k]).reshape(ref.shape), ref, atol=5e-5), k
            assert float(sd[k].norm()) == pytest.approx(last["paka_state_norm"][k], rel=1e-5), k


def test_dinov2_checkpoint_loads_without_the_paka_heads():
    """The post-training start (dinov31.py:180-205): a DINOv2 state_dict has no PaKA keys; they keep their initial values."""
    fx = torch.load(os.path.join(GOLD, "dinov31_d64.pt"), weights_only=False)
    with ops_emu.emulate(ops):
        m = build(fx)
        sd = {k: v for k, v in m.state_dict().items() if "_paka_head." not in k}
        before = m.student.p["paka.2.weight"].clone()
        m.load_state_dict(sd, strict=True)
        assert torch.equal(m.student.p["paka.2.weight"], before)
        with pytest.raises(KeyError):
            m.load_state_dict({k: v for k, v in sd.items() if "cls_token" not in k}, strict=True)


def test_resume_keeps_the_paka_heads_own_adam_step_count():
    """torch.optim.AdamW counts steps per parameter and skips parameters without a gradient: with `paka_start_step = 1` the PaKA head has
    taken one Adam step less than everything else.  A checkpoint written after two steps carries both counts (state "step" 2 / 1), a fresh
    object resumed from it takes a third step BITWISE equal to the uninterrupted run's (same exact-arithmetic kernels), and a checkpoint
    written before the PaKA head's first step holds no state for it -- as torch writes none -- and loads."""
    fx = torch.load(os.path.join(GOLD, "dinov31_d64.pt"), weights_only=False)
    with ops_emu.emulate(ops):
        m = build(fx)
        exactify(m)

        def step(obj, si):
            rec = fx["steps"][si]
            obj.training_step_impl({"views": synth_views(fx, rec["seed"]), "geometries": rec["geometries"]}, si, masks=rec["masks"])
            obj.optimizer_step()
            obj.on_train_batch_end()

        step(m, 0)
        early = m.checkpoint_dict()
        names = [n for g in m.optimizer_state_dict()["param_groups"] for n in [g["name"]]]
        assert m.opt_step == 1 and m.paka_opt_steps == 0 and names
        osd = early["optimizer_states"][0]
        n_state = len(osd["state"])
        n_params = sum(len(g["params"]) for g in osd["param_groups"])
        assert n_params - n_state == 6, "the six PaKA tensors have no optimizer state before their first step"
        step(m, 1)
        ck = m.checkpoint_dict()
        steps = sorted({int(float(st["step"])) for st in ck["optimizer_states"][0]["state"].values()})
        assert steps == [1, 2] and m.paka_opt_steps == 1
        m._pending.clear()     # (the pending batch-center sums are no part of a checkpoint, in the reference either: dinov2_loss.py:139-160)
        step(m, 2)

        r = build(fx)
        exactify(r)
        r.load_checkpoint_dict(ck)
        assert r.opt_step == 2 and r.paka_opt_steps == 1
        step(r, 2)
        assert torch.equal(r.student.data, m.student.data) and torch.equal(r.exp_avg, m.exp_avg) and torch.equal(r.exp_avg_sq, m.exp_avg_sq)

        e = build(fx)
        exactify(e)
        e.load_checkpoint_dict(early)
        assert e.opt_step == 1 and e.paka_opt_steps == 0
        p0, p1 = e._paka_span
        assert float(e.exp_avg[p0:p1].abs().max()) == 0.0
