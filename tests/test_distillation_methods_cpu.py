"""CPU: the ORCHESTRATION of the distillation methods (lightly_train_amd/distillationv3.py, distillation.py: frozen-teacher pass, mixup,
projection heads, token resampling between grids, the queue, KL / MSE losses, explicit backward through ViT / DINOv3 / ResNet students,
AdamW and LARS with the generic schedule) in exact arithmetic -- plain-torch stand-ins for the HIP ops (tests/tools/ops_emu.py), fp32
buffers -- against the fixtures the REFERENCE'S OWN CLASSES wrote (tests/golden/distill_*.pt): per step the loss terms and the gradient
norm, after the last step the parameters, the BatchNorm buffers and the queue, all to fp32 round-off.  The bf16 GPU runs of the same
fixtures (tests/test_gpu_distill.py, test_gpu_distill12.py) allow 1e-2 on the losses and ask for 95 % of the updates within 0.15 lr."""
import contextlib
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))

import lightly_train_amd  # noqa: E402,F401
import ops_emu  # noqa: E402
from lightly_train_amd import ops  # noqa: E402
from lightly_train_amd.vit import ViTConfig, Workspace  # noqa: E402


class F32Workspace(Workspace):
    def get(self, name, shape, dtype, **kw):
        return super().get(name, shape, torch.float32 if dtype == torch.bfloat16 else dtype, **kw)


class _NoStream:
    def __init__(self, *a, **k):
        pass

    def record_event(self):
        return None

    def wait_event(self, ev):
        pass

    def wait_stream(self, s):
        pass


@pytest.fixture(autouse=True)
def _no_cuda_streams(monkeypatch):
    monkeypatch.setattr(torch.cuda, "current_stream", lambda *a, **k: _NoStream())
    monkeypatch.setattr(torch.cuda, "set_stream", lambda s: None)
    monkeypatch.setattr(torch.cuda, "Stream", _NoStream)
    monkeypatch.setattr(torch.cuda, "stream", lambda s: contextlib.nullcontext())


def exactify(m):
    """Every bf16 buffer of the method becomes fp32: weight shadows, derived weight copies, activations."""
    m.ws = F32Workspace(torch.device("cpu"))
    for fp in (m.student, m.teacher):
        fp.bf16 = fp.data.clone()
        fp.b = {n: fp.bf16[fp.offsets[n]:fp.offsets[n] + fp.p[n].numel()].view(fp.shapes[n]) for n in fp.names}
    engines = [getattr(m, "s_vit", None), getattr(m, "t_vit", None), getattr(getattr(m, "s", None), "net", None), getattr(m, "s_net", None)]
    for e in engines:
        if e is None:
            continue
        if getattr(e, "wpe_pad", None) is not None:
            e.wpe_pad = e.wpe_pad.float()
        if hasattr(e, "w_stem"):                    # convolutional student
            e.act_dtype = torch.float32
            e.w_stem = e.w_stem.float()
        e.refresh_padded_weights()
    return m


def vit_cfg(c):
    return ViTConfig(embed_dim=c["embed_dim"], depth=c["depth"], num_heads=c["num_heads"], mlp_ratio=4.0, patch_size=c["patch_size"],
                     img_size=c["img_size"], init_values=c["init_values"])


def check_final(sd, fx, key, atol=3e-6):
    fin = fx["final"]
    conv = fx.get("student_cfg", {}).get("kind") == "resnet"
    for k, v in fin["student_backbone"].items():
        if conv and k.startswith("fc."):
            continue
        if k.endswith("num_batches_tracked"):
            assert int(sd[key + k]) == int(v), k
            continue
        assert torch.allclose(sd[key + k], v, atol=atol), (k, (sd[key + k] - v).abs().max().item())


@pytest.mark.parametrize("name", ["distill_v1_d64", "distill_v2_d64", "distill_v1_d64_lars", "distill_v2_d64_mlp3"])
def test_distillation_v1_v2_reproduce_the_reference_fixture(name):
    from lightly_train_amd.distillation import Distillation, DistillationArgs, DistillationV2, DistillationV2Args
    from lightly_train_amd.lars import LARSArgs

    fx = torch.load(os.path.join(GOLD, name + ".pt"), weights_only=False)
    with ops_emu.emulate(ops):
        kw = dict(global_batch_size=fx["b"], total_steps=fx["total_steps"], max_epochs=1, device="cpu", student_state=fx["init"]["student_backbone"],
                  teacher_state=fx["teacher_state"], head_state=fx["init"]["head"])
        scfg, tcfg = vit_cfg(fx["student_cfg"]), vit_cfg(fx["teacher_cfg"])
        if fx.get("optimizer") == "lars":
            m = Distillation(scfg, tcfg, DistillationArgs(queue_size=fx["queue_size"], optimizer="lars",
                                                          lars=LARSArgs(lr=fx["lr"], weight_decay=fx["weight_decay"])), **kw)
        elif fx["kind"] == "v1":
            m = Distillation(scfg, tcfg, DistillationArgs(queue_size=fx["queue_size"], optimizer="adamw", lr=fx["lr"], weight_decay=fx["weight_decay"]), **kw)
        else:
            m = DistillationV2(scfg, tcfg, DistillationV2Args(optimizer="adamw", lr=fx["lr"], weight_decay=fx["weight_decay"], n_projection_layers=fx.get("n_projection_layers", 1),
                                                          projection_hidden_dim=fx.get("projection_hidden_dim", 2048)), **kw)
        exactify(m)
        for si, rec in enumerate(fx["steps"]):
            x = torch.randn(fx["b"], 3, fx["img"], fx["img"], generator=torch.Generator().manual_seed(rec["x_seed"]))
            torch.manual_seed(400 + si)
            res = m.training_step_impl({"views": [x]}, 0)
            assert m._last["lam"] == pytest.approx(rec["lam"]) and torch.equal(m._last["index"], rec["index"])
            # LARS moves the no-decay tensors by plain SGD at lr 0.065 .. 0.13 on unit-norm gradients: the loss falls from 0.81 to 1e-3 within
            # a step and fp32 round-off is amplified accordingly (still two orders of magnitude tighter than the bf16 comparison)
            lars = fx.get("optimizer") == "lars"
            assert float(res.loss) == pytest.approx(rec["logs"]["loss"], rel=1e-3 if (lars and si > 0) else 5e-5, abs=1e-7), si
            m.optimizer_step()
            assert float(m.last_grad_norm.sqrt()) == pytest.approx(rec["logs"]["grad_norm"], rel=1e-2 if (lars and si > 0) else 2e-4), si
        sd = m.state_dict()
        check_final(sd, fx, "student_embedding_model.wrapped_model._model.", atol=2e-4 if lars else 3e-6)
        for k, v in fx["final"]["head"].items():
            assert torch.allclose(sd["student_projection_head." + k], v, atol=2e-4 if lars else 3e-6), k
        if fx["kind"] == "v1":
            assert torch.allclose(m.teacher_queue, fx["final"]["queue"], atol=2e-6)


@pytest.mark.parametrize("name", ["distill_v3_d64", "distill_v3_d64_p14", "distill_v3_d64_v3s", "distill_v3_resnet"])
def test_distillation_v3_reproduces_the_reference_fixture(name):
    from lightly_train_amd.dinov3 import convert_dinov3_state, dinov3_vit_config
    from lightly_train_amd.distillationv3 import DistillationV3, DistillationV3Args

    fx = torch.load(os.path.join(GOLD, name + ".pt"), weights_only=False)
    sc, tc = fx["student_cfg"], fx["teacher_cfg"]
    with ops_emu.emulate(ops):
        student_state = fx["init"]["student_backbone"]
        if sc.get("kind") == "resnet":
            from lightly_train_amd.resnet import ResNetConfig
            scfg = ResNetConfig(layers=tuple(sc["layers"]), width=sc["width"])
        elif sc.get("kind") == "dinov3":
            scfg = dinov3_vit_config(sc["embed_dim"], sc["depth"], sc["num_heads"], patch_size=sc["patch_size"], img_size=sc["img_size"],
                                     n_storage_tokens=sc["n_storage_tokens"], layerscale_init=sc["init_values"], rope_base=sc["rope_base"],
                                     ln_eps=sc["ln_eps"], rope_rescale=sc["rope_rescale"])
            student_state = convert_dinov3_state(student_state, scfg)
        else:
            scfg = vit_cfg(sc)
        tcfg = dinov3_vit_config(tc["embed_dim"], tc["depth"], tc["num_heads"], patch_size=tc["patch_size"], img_size=tc["img_size"],
                                 n_storage_tokens=tc["n_storage_tokens"], layerscale_init=0.5, rope_base=tc["rope_base"], ln_eps=tc["ln_eps"])
        m = DistillationV3(scfg, tcfg, DistillationV3Args(queue_size=fx["queue_size"], weight_decay=fx["weight_decay"]), global_batch_size=fx["b"],
                           total_steps=fx["total_steps"], max_epochs=1, device="cpu", student_state=student_state,
                           teacher_state=convert_dinov3_state(fx["teacher_state"], tcfg), proj_global_state=fx["init"]["proj_global"],
                           proj_local_state=fx["init"]["proj_local"])
        exactify(m)
        img = fx.get("img", 64)
        for si, rec in enumerate(fx["steps"]):
            x = torch.randn(fx["b"], 3, img, img, generator=torch.Generator().manual_seed(rec["x_seed"]))
            torch.manual_seed(300 + si)
            res = m.training_step_impl({"views": [x]}, 0)
            assert m._last["lam"] == pytest.approx(rec["lam"]) and torch.equal(m._last["index"], rec["index"])
            logs = {k.split("/")[-1]: float(v) for k, v in res.log_dict.items()}
            assert logs["global_loss"] == pytest.approx(rec["logs"]["global_loss"], rel=5e-5, abs=1e-7), si
            assert logs["local_loss"] == pytest.approx(rec["logs"]["local_loss"], rel=5e-4, abs=1e-7), si
            m.optimizer_step()
            assert float(m.last_grad_norm.sqrt()) == pytest.approx(rec["logs"]["grad_norm"], rel=2e-4), si
        assert torch.allclose(m.teacher_queue, fx["final"]["queue"], atol=2e-6)
        sd = m.state_dict()
        conv = sc.get("kind") == "resnet"
        check_final(sd, fx, "student_embedding_model.wrapped_model." + ("_features." if conv else "_model."))


def _build_any(name, fresh=False):
    """The method object of a fixture; fresh=True: from a random state instead of the fixture's initial one (a resume target)."""
    from lightly_train_amd.dinov3 import convert_dinov3_state, dinov3_vit_config
    from lightly_train_amd.distillation import Distillation, DistillationArgs, DistillationV2, DistillationV2Args
    from lightly_train_amd.distillationv3 import DistillationV3, DistillationV3Args
    from lightly_train_amd.lars import LARSArgs

    fx = torch.load(os.path.join(GOLD, name + ".pt"), weights_only=False)
    common = dict(global_batch_size=fx["b"], total_steps=fx["total_steps"], max_epochs=1, device="cpu", seed=77 if fresh else 0)
    if "kind" in fx:      # v1 / v2
        kw = dict(common, student_state=None if fresh else fx["init"]["student_backbone"], teacher_state=fx["teacher_state"],
                  head_state=None if fresh else fx["init"]["head"])
        scfg, tcfg = vit_cfg(fx["student_cfg"]), vit_cfg(fx["teacher_cfg"])
        if fx.get("optimizer") == "lars":
            m = Distillation(scfg, tcfg, DistillationArgs(queue_size=fx["queue_size"], optimizer="lars", lars=LARSArgs(lr=fx["lr"], weight_decay=fx["weight_decay"])), **kw)
        elif fx["kind"] == "v1":
            m = Distillation(scfg, tcfg, DistillationArgs(queue_size=fx["queue_size"], optimizer="adamw", lr=fx["lr"], weight_decay=fx["weight_decay"]), **kw)
        else:
            m = DistillationV2(scfg, tcfg, DistillationV2Args(optimizer="adamw", lr=fx["lr"], weight_decay=fx["weight_decay"], n_projection_layers=fx.get("n_projection_layers", 1),
                                                          projection_hidden_dim=fx.get("projection_hidden_dim", 2048)), **kw)
        return fx, exactify(m), 400
    sc, tc = fx["student_cfg"], fx["teacher_cfg"]
    student_state = fx["init"]["student_backbone"]
    if sc.get("kind") == "resnet":
        from lightly_train_amd.resnet import ResNetConfig
        scfg = ResNetConfig(layers=tuple(sc["layers"]), width=sc["width"])
    elif sc.get("kind") == "dinov3":
        scfg = dinov3_vit_config(sc["embed_dim"], sc["depth"], sc["num_heads"], patch_size=sc["patch_size"], img_size=sc["img_size"],
                                 n_storage_tokens=sc["n_storage_tokens"], layerscale_init=sc["init_values"], rope_base=sc["rope_base"], ln_eps=sc["ln_eps"],
                                 rope_rescale=sc["rope_rescale"])
        student_state = convert_dinov3_state(student_state, scfg)
    else:
        scfg = vit_cfg(sc)
    tcfg = dinov3_vit_config(tc["embed_dim"], tc["depth"], tc["num_heads"], patch_size=tc["patch_size"], img_size=tc["img_size"],
                             n_storage_tokens=tc["n_storage_tokens"], layerscale_init=0.5, rope_base=tc["rope_base"], ln_eps=tc["ln_eps"])
    m = DistillationV3(scfg, tcfg, DistillationV3Args(queue_size=fx["queue_size"], weight_decay=fx["weight_decay"]), teacher_state=convert_dinov3_state(fx["teacher_state"], tcfg),
                       student_state=None if fresh else student_state, proj_global_state=None if fresh else fx["init"]["proj_global"],
                       proj_local_state=None if fresh else fx["init"]["proj_local"], **common)
    return fx, exactify(m), 300


@pytest.mark.parametrize("name", ["distill_v1_d64", "distill_v1_d64_lars", "distill_v2_d64", "distill_v3_d64", "distill_v3_d64_v3s", "distill_v3_resnet"])
def test_resume_of_the_distillation_methods_is_exact(name):
    """f4 for the sibling methods: state_dict() + optimizer_state() after two steps, loaded into an object built from a DIFFERENT random
    state, give the third step of the uninterrupted run bit for bit (parameters, BatchNorm buffers, queue)."""
    with ops_emu.emulate(ops):
        fx, a, seed0 = _build_any(name)
        img = fx.get("img", 64)

        def step(m, si):
            x = torch.randn(fx["b"], 3, img, img, generator=torch.Generator().manual_seed(fx["steps"][si]["x_seed"]))
            torch.manual_seed(seed0 + si)
            m.training_step_impl({"views": [x]}, 0)
            m.optimizer_step()

        for si in range(2):
            step(a, si)
        sd, ost = a.state_dict(), a.optimizer_state()
        step(a, 2)
        _, b, _ = _build_any(name, fresh=True)
        assert any(not torch.equal(v, b.state_dict()[k]) for k, v in sd.items())          # really a different object
        b.load_state_dict(sd)
        b.load_optimizer_state(ost)
        assert all(torch.equal(v, b.state_dict()[k]) for k, v in sd.items())
        step(b, 2)
        fa, fb = a.state_dict(), b.state_dict()
        assert list(fa) == list(fb)
        for k in fa:
            assert torch.equal(fa[k], fb[k]), k
        with pytest.raises(KeyError):
            b.load_state_dict({k: v for k, v in sd.items() if "projection_head" not in k})


@pytest.mark.parametrize("name,b,queue,tg,tl,wl,img", [
    ("distill_v3_d64", 3, 16, 0.07, 0.07, 1.0, 64),
    ("distill_v3_d64", 40, 32, 0.05, 0.1, 0.5, 64),        # batch >= queue: the queue becomes the first Q teacher features (distillationv3.py:275-291)
    ("distill_v3_d64", 5, 32, 0.1, 0.04, 2.0, 96),         # student positional embedding resampled to a 6x6 grid
    ("distill_v3_d64_p14", 4, 8, 0.07, 0.07, 1.0, 112),    # 8x8 student grid onto the teacher's 7x7
    ("distill_v3_d64_v3s", 6, 32, 0.07, 0.2, 1.5, 64),     # DINOv3 student, per-block RoPE rescale draws
    ("distill_v3_resnet", 16, 16, 0.07, 0.07, 0.25, 64),   # ResNet student, batch == queue
])
def test_drawn_distillationv3_configurations_equal_the_restatement(name, b, queue, tg, tl, wl, img):
    """Batch / queue sizes on both sides of the queue-replacement rule, temperatures, the local-loss weight and image sizes that move the
    student and teacher grids apart, on the four student kinds: three steps against the pinned restatement (loss terms 5e-5, gradient norm
    2e-4, parameters after the steps within the Adam-sign bound)."""
    from lightly_train_amd.dinov3 import convert_dinov3_state, dinov3_vit_config
    from lightly_train_amd.distillationv3 import DistillationV3, DistillationV3Args
    from oracle import distill_oracle as OD

    fx = torch.load(os.path.join(GOLD, name + ".pt"), weights_only=False)
    sc, tc = dict(fx["student_cfg"]), dict(fx["teacher_cfg"])
    with ops_emu.emulate(ops):
        student_state = fx["init"]["student_backbone"]
        if sc.get("kind") == "resnet":
            from lightly_train_amd.resnet import ResNetConfig
            scfg = ResNetConfig(layers=tuple(sc["layers"]), width=sc["width"])
        elif sc.get("kind") == "dinov3":
            scfg = dinov3_vit_config(sc["embed_dim"], sc["depth"], sc["num_heads"], patch_size=sc["patch_size"], img_size=sc["img_size"],
                                     n_storage_tokens=sc["n_storage_tokens"], layerscale_init=sc["init_values"], rope_base=sc["rope_base"],
                                     ln_eps=sc["ln_eps"], rope_rescale=sc["rope_rescale"])
            student_state = convert_dinov3_state(student_state, scfg)
        else:
            scfg = vit_cfg(sc)
        tcfg = dinov3_vit_config(tc["embed_dim"], tc["depth"], tc["num_heads"], patch_size=tc["patch_size"], img_size=tc["img_size"],
                                 n_storage_tokens=tc["n_storage_tokens"], layerscale_init=0.5, rope_base=tc["rope_base"], ln_eps=tc["ln_eps"])
        args = DistillationV3Args(queue_size=queue, weight_decay=fx["weight_decay"], temperature_global=tg, temperature_local=tl, loss_local_weight=wl)
        m = exactify(DistillationV3(scfg, tcfg, args, global_batch_size=b, total_steps=30, max_epochs=1, device="cpu", student_state=student_state,
                                    teacher_state=convert_dinov3_state(fx["teacher_state"], tcfg), proj_global_state=fx["init"]["proj_global"],
                                    proj_local_state=fx["init"]["proj_local"]))
        o = OD.OracleDistillationV3(fx["init"]["student_backbone"], sc, fx["teacher_state"], tc, fx["init"]["proj_global"], fx["init"]["proj_local"], queue, b, 30,
                                    temperature_global=tg, temperature_local=tl, loss_local_weight=wl, weight_decay=fx["weight_decay"])
        g = torch.Generator().manual_seed(b * 1000 + queue)
        for si in range(3):
            x = torch.randn(b, 3, img, img, generator=g)
            torch.manual_seed(40 + si)
            res = m.training_step_impl({"views": [x]}, 0)
            torch.manual_seed(40 + si)
            lam = torch.empty(1).uniform_(0.0, 1.0).item()
            index = torch.randperm(b)
            assert m._last["lam"] == pytest.approx(lam) and torch.equal(m._last["index"], index)
            ol = o.train_step(x, lam, index)            # its RoPE rescale draws follow the mixup draws from the same generator state
            logs = {k.split("/")[-1]: float(v) for k, v in res.log_dict.items()}
            assert logs["global_loss"] == pytest.approx(ol["global_loss"], rel=5e-5, abs=1e-7), si
            assert logs["local_loss"] == pytest.approx(ol["local_loss"], rel=5e-4, abs=1e-7), si
            m.optimizer_step()
            assert float(m.last_grad_norm.sqrt()) == pytest.approx(ol["grad_norm"], rel=2e-4), si
        assert torch.allclose(m.teacher_queue, o.queue, atol=2e-6)
        assert torch.allclose(m.student.p["proj_global.weight"], o.pg["weight"].detach(), atol=3e-6)
        assert torch.allclose(m.student.p["proj_local.weight"], o.pl["weight"].detach(), atol=3e-6)


def test_distillation_state_dict_chunked_names_and_validate_before_copy():
    """A chunked student (block_chunks > 0, the reference's vitl14 / vitg14 layout) writes and reads `blocks.<chunk>.<i>.` names; a load
    that fails -- unknown key, wrong shape, missing tensor -- leaves the model untouched; strict=False tolerates unknown / missing keys."""
    from lightly_train_amd.distillation import Distillation, DistillationArgs

    fx = torch.load(os.path.join(GOLD, "distill_v1_d64.pt"), weights_only=False)
    scfg, tcfg = vit_cfg(fx["student_cfg"]), vit_cfg(fx["teacher_cfg"])
    scfg.block_chunks = 2
    with ops_emu.emulate(ops):
        kw = dict(global_batch_size=fx["b"], total_steps=fx["total_steps"], max_epochs=1, device="cpu", teacher_state=fx["teacher_state"])
        a = Distillation(scfg, tcfg, DistillationArgs(queue_size=fx["queue_size"], optimizer="adamw"), seed=1, **kw)
        b = Distillation(scfg, tcfg, DistillationArgs(queue_size=fx["queue_size"], optimizer="adamw"), seed=2, **kw)
        sd = a.state_dict()
        pre = "student_embedding_model.wrapped_model._model."
        assert pre + "blocks.0.0.attn.qkv.weight" in sd and pre + "blocks.1.1.mlp.fc2.bias" in sd and not any(k.startswith(pre + "blocks.0.attn") for k in sd)
        before = b.student.data.clone()
        assert not torch.equal(before, a.student.data)
        bad_key = dict(sd, **{"student_projection_head.nonsense.weight": torch.zeros(3)})
        with pytest.raises(KeyError):
            b.load_state_dict(bad_key)
        bad_shape = dict(sd)
        last = [k for k in sd if k.startswith("student_projection_head.")][-1]
        bad_shape[last] = torch.zeros(5, 7)
        with pytest.raises(ValueError):
            b.load_state_dict(bad_shape)
        missing = {k: v for k, v in sd.items() if "blocks.1.1.mlp.fc2.bias" not in k}
        with pytest.raises(KeyError):
            b.load_state_dict(missing)
        assert torch.equal(b.student.data, before), "a failed load must not have copied anything"
        b.load_state_dict(bad_key, strict=False)          # unknown keys tolerated
        assert torch.equal(b.student.data, a.student.data)
        b.student.data.copy_(before)
        b.load_state_dict(missing, strict=False)          # missing tensors keep their values
        name = "backbone.blocks.1.mlp.fc2.bias"
        o, n = b.student.offsets[name], b.student.p[name].numel()
        assert torch.equal(b.student.data[o:o + n], before[o:o + n])
        b.load_state_dict(sd)
        assert torch.equal(b.student.data, a.student.data) and torch.equal(b.teacher_queue, a.teacher_queue)


def test_optimizer_state_load_zeroes_moments_without_an_entry():
    """checkpoint.load_optimizer_state_dict: a step-0 (empty-state) checkpoint loaded into an object that has trained resets the Adam
    moments, as torch's Optimizer.load_state_dict drops state it is not given."""
    from lightly_train_amd.dinov2 import DINOv2, DINOv2Args

    with ops_emu.emulate(ops):
        cfg = ViTConfig(embed_dim=64, depth=1, num_heads=1, mlp_ratio=4.0, patch_size=16, img_size=32, init_values=0.1)
        m = DINOv2(cfg, DINOv2Args(output_dim=64, hidden_dim=32, dino_bottleneck_dim=16), global_batch_size=2, total_steps=10, device="cpu")
        fresh = m.optimizer_state_dict()
        assert fresh["state"] == {}
        m.exp_avg.fill_(3.0); m.exp_avg_sq.fill_(2.0); m.opt_step = 7
        m.load_optimizer_state_dict(fresh)
        assert m.opt_step == 0 and float(m.exp_avg.abs().max()) == 0.0 and float(m.exp_avg_sq.abs().max()) == 0.0


def test_dinov3_teacher_state_with_untied_local_cls_norm_loads_and_matches_the_reference_in_eval():
    """`untie_global_and_local_cls_norm=True` (the SAT-493M ViT-L / ViT-7B recipes): `local_cls_norm` is read by training-mode
    local-crop forwards only (vision_transformer.py:286-292), never by a frozen eval() teacher -- the converted state drops it and the
    engine's eval forward equals the reference model's."""
    from oracle import ref_harness as H

    if not H.reference_available():
        pytest.skip("/root/reference not present (GPU box)")
    H.install()
    from lightly_train._models.dinov3.dinov3_src.models.vision_transformer import DinoVisionTransformer as V3
    from lightly_train_amd.dinov3 import convert_dinov3_state, dinov3_vit_config
    from lightly_train_amd.params import FlatParams
    from lightly_train_amd.vit import ViTEngine, vit_param_shapes

    torch.manual_seed(5)
    ref = V3(img_size=64, patch_size=16, embed_dim=64, depth=2, num_heads=1, ffn_ratio=4, qkv_bias=True, layerscale_init=0.1, norm_layer="layernormbf16",
             ffn_layer="mlp", n_storage_tokens=4, mask_k_bias=True, pos_embed_rope_base=100, pos_embed_rope_normalize_coords="separate",
             pos_embed_rope_rescale_coords=None, pos_embed_rope_dtype="fp32", untie_global_and_local_cls_norm=True)
    ref.init_weights()
    with torch.no_grad():
        for n_, p_ in ref.named_parameters():
            if "norm" in n_ or n_.endswith("bias"):
                p_.add_(0.1 * torch.randn_like(p_))
    ref.eval()
    sd = ref.state_dict()
    assert any(k.startswith("local_cls_norm.") for k in sd)
    cfg = dinov3_vit_config(64, 2, 1, patch_size=16, img_size=64)
    conv = convert_dinov3_state(sd, cfg)
    assert not any(k.startswith("local_cls_norm.") for k in conv)
    x = torch.randn(3, 3, 64, 64, generator=torch.Generator().manual_seed(1))
    with ops_emu.emulate(ops):
        fp = FlatParams([(n, conv[n]) for n, _ in vit_param_shapes(cfg)], "cpu", False)
        fp.bf16 = fp.data.clone()
        fp.b = {n: fp.bf16[fp.offsets[n]:fp.offsets[n] + fp.p[n].numel()].view(fp.shapes[n]) for n in fp.names}
        eng = ViTEngine(cfg, fp, "")
        if eng.wpe_pad is not None:
            eng.wpe_pad = eng.wpe_pad.float()
        eng.refresh_padded_weights()
        ctx = eng.forward(F32Workspace(torch.device("cpu")), "t", x, None, save=False)
    with torch.no_grad():
        out = ref.forward_features(x)
    xn = ctx["xn"].view(3, -1, 64)
    assert torch.allclose(xn[:, 0], out["x_norm_clstoken"], atol=2e-5) and torch.allclose(xn[:, 5:], out["x_norm_patchtokens"], atol=2e-5)


def test_distillation_v1_queue_size_off_the_8_grid():
    """A queue size that is not a multiple of 8 (the similarity rows are padded to 8 columns inside, lt_kl_fwd_bwd reads the real ones):
    three steps against the restated method (oracle/distill_oracle.py, itself pinned on the reference's fixture at queue 32), through the
    queue filling up and rolling (37 entries, batch 8)."""
    from lightly_train_amd.distillation import Distillation, DistillationArgs
    from oracle import distill_oracle as OD

    fx = torch.load(os.path.join(GOLD, "distill_v1_d64.pt"), weights_only=False)
    Q = 37
    o = OD.OracleDistillation12("v1", fx["init"]["student_backbone"], fx["student_cfg"], fx["teacher_state"], fx["teacher_cfg"], fx["init"]["head"], Q,
                                fx["b"], fx["total_steps"], lr=fx["lr"], weight_decay=fx["weight_decay"])
    with ops_emu.emulate(ops):
        m = exactify(Distillation(vit_cfg(fx["student_cfg"]), vit_cfg(fx["teacher_cfg"]),
                                  DistillationArgs(queue_size=Q, optimizer="adamw", lr=fx["lr"], weight_decay=fx["weight_decay"]),
                                  global_batch_size=fx["b"], total_steps=fx["total_steps"], max_epochs=1, device="cpu",
                                  student_state=fx["init"]["student_backbone"], teacher_state=fx["teacher_state"], head_state=fx["init"]["head"]))
        for s in range(6):
            x = torch.randn(fx["b"], 3, fx["img"], fx["img"], generator=torch.Generator().manual_seed(900 + s))
            lam, index = 0.3 + 0.1 * s, torch.randperm(fx["b"], generator=torch.Generator().manual_seed(s))
            want = o.train_step(x, lam, index)
            res = m.train_step(x, mix=(lam, index))
            assert float(res.loss) == pytest.approx(want["loss"], rel=2e-5, abs=1e-7), s
            assert m._last["t_logits"].shape == (fx["b"], Q)
        assert torch.allclose(m.teacher_queue, o.queue, atol=1e-6)
        for k, v in o.sb.items():
            assert torch.allclose(m.student.p["backbone." + k], v.detach(), atol=3e-6), k
