"""CPU: the ORCHESTRATION of the DINO (v1) step (lightly_train_amd/dino.py: EMA before the forward, pooled class tokens through one
projection head per role, lightly's DINOLoss pairing and normalisation, immediate center update, last-layer freeze at lr 0 / wd 0 with
moving optimizer moments, weight-decay grouping and schedule, SGD / AdamW, clipping at 3.0) in exact arithmetic -- plain-torch stand-ins
for the HIP ops (tests/tools/ops_emu.py), fp32 buffers -- against the fixtures the REFERENCE'S OWN `DINO` class wrote
(tests/golden/dino_v1_d64*.pt; the LightlySSL loss / head / grouping inside it are restated, oracle/dino_oracle.py): per step the loss,
the gradient norm, the schedules and the center; the first step's gradients tensor by tensor; after four steps every parameter of
student and teacher and the optimizer's state, to fp32 round-off.  The bf16 GPU run of the same fixtures: tests/test_gpu_dino_v1.py."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import lightly_train_amd  # noqa: E402,F401
import ops_emu  # noqa: E402
from lightly_train_amd import ops  # noqa: E402
from lightly_train_amd.dino import DINO, DINOArgs, decays, teacher_temp_schedule  # noqa: E402
from test_distillation_methods_cpu import _no_cuda_streams, exactify, vit_cfg  # noqa: E402,F401


def exact(m):
    """test_distillation_methods_cpu.exactify + the heads' derived (weight-normed) prototype matrices in fp32."""
    exactify(m)
    for h in (m.s_head, m.t_head):
        h.wn = h.wn.float()
    m._refresh_derived()
    return m


def build(fx, device="cpu"):
    ma = fx["method_args"]
    oa = fx["optimizer_args"]
    args = DINOArgs(hidden_dim=ma["hidden_dim"], bottleneck_dim=ma["bottleneck_dim"], output_dim=ma["output_dim"],
                    student_freeze_last_layer_steps=ma["student_freeze_last_layer_steps"], norm_last_layer=ma["norm_last_layer"],
                    teacher_temp=ma["teacher_temp"], warmup_teacher_temp=ma["warmup_teacher_temp"], warmup_teacher_temp_steps=ma["warmup_teacher_temp_steps"],
                    student_temp=ma["student_temp"], center_momentum=ma["center_momentum"], momentum_start=ma["momentum_start"], momentum_end=ma["momentum_end"],
                    weight_decay_start=ma["weight_decay_start"], weight_decay_end=ma["weight_decay_end"], warmup_steps=ma["warmup_steps"],
                    optimizer=fx["optimizer"], lr=oa["lr"], weight_decay=oa["weight_decay"])
    init = fx["init"]
    return DINO(vit_cfg(fx["cfg"]), args, global_batch_size=fx["b"], total_steps=fx["total_steps"], device=device, backbone_state=init["student_backbone"],
                teacher_backbone_state=init["teacher_backbone"], student_head_state=init["student_head"], teacher_head_state=init["teacher_head"])


def views_of(fx, rec):
    g = torch.Generator().manual_seed(rec["view_seed"])
    return [torch.randn(fx["b"], 3, fx["g_size"], fx["g_size"], generator=g) for _ in range(2)] + \
           [torch.randn(fx["b"], 3, fx["l_size"], fx["l_size"], generator=g) for _ in range(fx["n_local"])]


@pytest.mark.parametrize("name", ["dino_v1_d64", "dino_v1_d64_adamw"])
def test_dino_v1_reproduces_the_reference_fixture(name):
    fx = torch.load(os.path.join(GOLD, name + ".pt"), weights_only=False)
    with ops_emu.emulate(ops):
        m = exact(build(fx))
        assert {g: m.groups.count(g) for g in set(m.groups)} == fx["groups"]
        for si, rec in enumerate(fx["steps"]):
            res = m.training_step_impl({"views": views_of(fx, rec)}, 0)
            logs = rec["logs"]
            assert float(res.loss) == pytest.approx(logs["loss"], rel=2e-6)
            assert res.log_dict["schedule/momentum"] == pytest.approx(logs["momentum"], rel=1e-12)
            assert res.log_dict["schedule/teacher_temp"] == pytest.approx(logs["teacher_temp"], rel=1e-12)
            B = fx["b"]
            sw = torch.cat([rec["teacher_logits"][B:], rec["teacher_logits"][:B]])     # our teacher rows are stored with the views swapped
            assert torch.allclose(m._last["t_logits"][:2 * B], sw, atol=2e-5)
            assert torch.allclose(m._last["s_global_logits"], rec["student_global_logits"], atol=2e-5)
            assert torch.allclose(m._last["s_local_logits"], rec["student_local_logits"], atol=2e-5)
            h = m._hparams_now()
            hp = rec["hparams"]
            assert m.base_lr * h["lr_factor"] == pytest.approx(hp["params"]["lr"], rel=1e-9)
            assert h["weight_decay"] == pytest.approx(hp["params"]["weight_decay"], rel=1e-9)
            assert h["frozen"] == (hp["params_last_layer"]["lr"] == 0.0 and hp["params_last_layer"]["weight_decay"] == 0.0)
            gn = float(torch.sqrt((m.student.grad.double() ** 2).sum()))
            assert gn == pytest.approx(logs["grad_norm"], rel=2e-5)
            if "grads" in rec:
                clip = min(1.0, 3.0 / (logs["grad_norm"] + 1e-6))
                for n in m.student.names:
                    key = m._ref_key("student", n)
                    if key in rec["no_grad"]:
                        assert float(m.student.g[n].abs().max()) == 0.0, n
                        continue
                    want = rec["grads"][key] / clip
                    assert torch.allclose(m.student.g[n], want, atol=1e-6 + 2e-5 * float(want.abs().max())), (n, (m.student.g[n] - want).abs().max())
            m.optimizer_step()
            assert torch.allclose(m.center.view(-1), rec["center"].view(-1), atol=1e-6)
        sd = m.state_dict()
        assert list(sd.keys()) == fx["state_dict_keys"]
        fin = fx["final"]
        for role in ("student", "teacher"):
            for k, v in fin[role + "_backbone"].items():
                got = sd[f"{role}_embedding_model.wrapped_model._model.{k}"]
                assert torch.allclose(got, v, atol=2e-6), (role, k, (got - v).abs().max())
            for k, v in fin[role + "_head"].items():
                got = sd[f"{role}_projection_head.{k}"]
                assert torch.allclose(got, v, atol=2e-6), (role, k, (got - v).abs().max())
        # the optimizer's own state in the reference's layout: groups, parameter order, momentum buffers / Adam moments
        osd, ref = m.optimizer_state_dict(), fx["optimizer_state"]
        assert [(g["name"], g["params"]) for g in osd["param_groups"]] == [(g["name"], g["params"]) for g in ref["param_groups"]]
        assert sorted(osd["state"]) == sorted(ref["state"])
        for i, st in ref["state"].items():
            for k, v in st.items():
                if torch.is_tensor(v) and v.ndim > 0:
                    assert torch.allclose(osd["state"][i][k], v, atol=1e-6 + 2e-5 * float(v.abs().max())), (i, k)


def test_dino_v1_resume_continues_the_same_trajectory():
    fx = torch.load(os.path.join(GOLD, "dino_v1_d64.pt"), weights_only=False)
    with ops_emu.emulate(ops):
        a, b = exact(build(fx)), exact(build(fx))
        for rec in fx["steps"][:2]:
            a.train_step(views_of(fx, rec))
        b.load_checkpoint_dict(a.checkpoint_dict())
        for rec in fx["steps"][2:]:
            ra, rb = a.train_step(views_of(fx, rec)), b.train_step(views_of(fx, rec))
            assert float(ra.loss) == float(rb.loss)
        assert torch.equal(a.student.data, b.student.data) and torch.equal(a.teacher.data, b.teacher.data)
        assert torch.equal(a.momentum_buffer, b.momentum_buffer) and torch.equal(a.center, b.center)


def test_dino_v1_host_rules():
    assert teacher_temp_schedule(0.07, 0.04, 3, 0) == 0.04 and teacher_temp_schedule(0.07, 0.04, 3, 3) == 0.07
    assert teacher_temp_schedule(0.07, 0.04, 3, 1) == pytest.approx(0.05)
    for n, want in (("backbone.cls_token", True), ("backbone.pos_embed", True), ("backbone.blocks.0.ls1.gamma", True), ("backbone.blocks.0.norm1.weight", False),
                    ("backbone.norm.weight", False), ("backbone.blocks.1.attn.qkv.bias", False), ("backbone.blocks.1.mlp.fc1.weight", True),
                    ("head.mlp.0.weight", True), ("head.mlp.0.bias", False)):
        assert decays(n) is want, n
    # resolve_auto of the reference (dino.py:80-196) at two dataset sizes
    small, big = DINOArgs.for_dataset_size(10_000), DINOArgs.for_dataset_size(2_000_000)
    assert (small.output_dim, small.teacher_temp, small.warmup_teacher_temp, small.momentum_start) == (1024, 0.02, 0.02, 0.99)
    assert (big.output_dim, big.teacher_temp, big.warmup_teacher_temp, big.momentum_start) == (65536, 0.07, 0.04, 0.996)
    with pytest.raises(ValueError):
        DINO(vit_cfg(dict(embed_dim=64, depth=1, num_heads=1, patch_size=16, img_size=32, init_values=0.1)), DINOArgs(optimizer="lamb"), device="cpu")


def build_resnet(fx, device="cpu"):
    from lightly_train_amd.dino import DINOResNet
    from lightly_train_amd.resnet import ResNetConfig

    ma, oa, c = fx["method_args"], fx["optimizer_args"], fx["cfg"]
    args = DINOArgs(hidden_dim=ma["hidden_dim"], bottleneck_dim=ma["bottleneck_dim"], output_dim=ma["output_dim"],
                    student_freeze_last_layer_steps=ma["student_freeze_last_layer_steps"], norm_last_layer=ma["norm_last_layer"],
                    teacher_temp=ma["teacher_temp"], warmup_teacher_temp=ma["warmup_teacher_temp"], warmup_teacher_temp_steps=ma["warmup_teacher_temp_steps"],
                    student_temp=ma["student_temp"], center_momentum=ma["center_momentum"], momentum_start=ma["momentum_start"], momentum_end=ma["momentum_end"],
                    weight_decay_start=ma["weight_decay_start"], weight_decay_end=ma["weight_decay_end"], warmup_steps=ma["warmup_steps"],
                    optimizer=fx["optimizer"], lr=oa["lr"], weight_decay=oa["weight_decay"])
    init = fx["init"]
    return DINOResNet(ResNetConfig(layers=tuple(c["layers"]), width=c["width"]), args, global_batch_size=fx["b"], total_steps=fx["total_steps"], device=device,
                      backbone_state=init["student_backbone"], teacher_backbone_state=init["teacher_backbone"], student_head_state=init["student_head"],
                      teacher_head_state=init["teacher_head"])


def exact_resnet(m):
    from test_distillation_methods_cpu import F32Workspace

    m.ws = F32Workspace(torch.device("cpu"))
    for fp in (m.student, m.teacher):
        fp.bf16 = fp.data.clone()
        fp.b = {n: fp.bf16[fp.offsets[n]:fp.offsets[n] + fp.p[n].numel()].view(fp.shapes[n]) for n in fp.names}
    for e in (m.s_net, m.t_net):
        e.act_dtype = torch.float32
        e.w_stem = e.w_stem.float()
        e.refresh_padded_weights()
    for h in (m.s_head, m.t_head):
        h.wn = h.wn.float()
        h.refresh_weightnorm()
    return m


def test_dino_v1_on_a_resnet_reproduces_the_reference_fixture():
    """`DINOResNet` (convolutional backbone: train-mode BatchNorm in student and teacher, two student forward calls, average-pooled
    features) in exact arithmetic against tests/golden/dino_v1_resnet.pt, written by the reference's `DINO` class around its
    `ResNetModelWrapper` (on the restated torchvision ResNet): per step loss, schedules, gradient norm, head outputs, center; the first
    step's gradients tensor by tensor; after four steps every parameter and BatchNorm buffer of student and teacher."""
    from lightly_train_amd.resnet import from_flat_layout

    fx = torch.load(os.path.join(GOLD, "dino_v1_resnet.pt"), weights_only=False)
    with ops_emu.emulate(ops):
        m = exact_resnet(build_resnet(fx))
        assert {g: m.groups.count(g) for g in set(m.groups)} == fx["groups"]
        B = fx["b"]
        for si, rec in enumerate(fx["steps"]):
            res = m.training_step_impl({"views": views_of(fx, rec)}, 0)
            logs = rec["logs"]
            assert float(res.loss) == pytest.approx(logs["loss"], rel=5e-6), si
            assert torch.allclose(m._last["t_logits"][:2 * B], rec["teacher_logits"], atol=5e-5)
            assert torch.allclose(m._last["s_global_logits"], rec["student_global_logits"], atol=5e-5)
            assert torch.allclose(m._last["s_local_logits"], rec["student_local_logits"], atol=5e-5)
            gn = float(torch.sqrt((m.student.grad.double() ** 2).sum()))
            assert gn == pytest.approx(logs["grad_norm"], rel=5e-5), si
            if "grads" in rec:
                clip = min(1.0, 3.0 / (logs["grad_norm"] + 1e-6))
                for n in m.student.names:
                    key = m._ref_key("student", n)
                    if key in rec["no_grad"]:
                        assert float(m.student.g[n].abs().max()) == 0.0, n
                        continue
                    want = rec["grads"][key] / clip
                    got = from_flat_layout(n[9:], m.student.g[n]) if n.startswith("backbone.") else m.student.g[n]
                    assert torch.allclose(got, want, atol=2e-6 + 1e-4 * float(want.abs().max())), (n, (got - want).abs().max())
            m.optimizer_step()
            assert torch.allclose(m.center.view(-1), rec["center"].view(-1), atol=1e-6)
        sd = m.state_dict()
        assert list(sd.keys()) == fx["state_dict_keys"]
        fin = fx["final"]
        for role in ("student", "teacher"):
            for k, v in fin[role + "_backbone"].items():
                got = sd[f"{role}_embedding_model.wrapped_model._features.{k}"]
                if k.endswith("num_batches_tracked"):
                    assert int(got) == int(v), (role, k)
                else:
                    assert torch.allclose(got, v, atol=5e-6), (role, k, (got - v).abs().max())
            for k, v in fin[role + "_head"].items():
                assert torch.allclose(sd[f"{role}_projection_head.{k}"], v, atol=5e-6), (role, k)
        # round trip through load_state_dict into an object built from another seed
        other = exact_resnet(build_resnet(dict(fx, init=dict(fx["init"], student_head=None, teacher_head=None))))
        other.load_state_dict(sd)
        assert torch.equal(other.student.data, m.student.data) and torch.equal(other.teacher.data, m.teacher.data)
        assert all(torch.equal(other.t_net.buffers[k], m.t_net.buffers[k]) for k in m.t_net.buffers)


def test_dino_v1_resnet_resume_continues_the_same_trajectory():
    """checkpoint_dict() / load_checkpoint_dict() of the convolutional class: parameters, both networks' BatchNorm buffers (the teacher's are
    its own: the EMA walks parameters only), the momentum buffers, the center and the step counters -- a second object built from another
    random state continues bit for bit."""
    fx = torch.load(os.path.join(GOLD, "dino_v1_resnet.pt"), weights_only=False)
    with ops_emu.emulate(ops):
        a = exact_resnet(build_resnet(fx))
        b = exact_resnet(build_resnet(dict(fx, init=dict(fx["init"], student_head=None, teacher_head=None))))
        for rec in fx["steps"][:2]:
            a.train_step(views_of(fx, rec))
        ck = a.checkpoint_dict()
        assert ck["global_step"] == 2 and len(ck["optimizer_states"][0]["state"]) == sum(1 for n in a.student.names if n not in a._untrained)
        b.load_checkpoint_dict(ck)
        for rec in fx["steps"][2:]:
            ra, rb = a.train_step(views_of(fx, rec)), b.train_step(views_of(fx, rec))
            assert float(ra.loss) == float(rb.loss)
        assert torch.equal(a.student.data, b.student.data) and torch.equal(a.teacher.data, b.teacher.data)
        assert torch.equal(a.momentum_buffer, b.momentum_buffer) and torch.equal(a.center, b.center)
        for net_a, net_b in ((a.s_net, b.s_net), (a.t_net, b.t_net)):
            assert all(torch.equal(net_a.buffers[k], net_b.buffers[k]) for k in net_a.buffers)
