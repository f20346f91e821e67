"""-m gpu: Distillation (v1) and DistillationV2 in HIP (lightly_train_amd/distillation.py) against
tests/golden/distill_v{1,2}_d64.pt, written by the reference's own classes on CPU (oracle/make_golden.py::make_distill12: frozen DINOv2
ViT teacher D=64 /14, DINOv2 ViT student /16, 112^2 images, AdamW, 3 steps), and against the oracle restatement for gradients.
Tolerances (bf16 MFMA operands vs fp32): loss 1e-2 relative, gradient norm 5e-2, gradients 5e-2 of max|grad| per tensor, after 3 AdamW
steps > 95 % of the parameter updates within 0.15 lr of the reference's."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-20)).item()


def build(fx):
    import lightly_train_amd  # noqa: F401
    from lightly_train_amd.distillation import Distillation, DistillationArgs, DistillationV2, DistillationV2Args
    from lightly_train_amd.vit import ViTConfig

    sc, tc = fx["student_cfg"], fx["teacher_cfg"]
    scfg = ViTConfig(embed_dim=sc["embed_dim"], depth=sc["depth"], num_heads=sc["num_heads"], mlp_ratio=4.0, patch_size=sc["patch_size"],
                     img_size=sc["img_size"], init_values=sc["init_values"])
    tcfg = ViTConfig(embed_dim=tc["embed_dim"], depth=tc["depth"], num_heads=tc["num_heads"], mlp_ratio=4.0, patch_size=tc["patch_size"],
                     img_size=tc["img_size"], init_values=tc["init_values"])
    kw = dict(global_batch_size=fx["b"], total_steps=fx["total_steps"], max_epochs=1, device="cuda", student_state=fx["init"]["student_backbone"],
              teacher_state=fx["teacher_state"], head_state=fx["init"]["head"])
    if fx.get("optimizer") == "lars":
        from lightly_train_amd.lars import LARSArgs
        return Distillation(scfg, tcfg, DistillationArgs(queue_size=fx["queue_size"], optimizer="lars",
                                                         lars=LARSArgs(lr=fx["lr"], weight_decay=fx["weight_decay"])), **kw)
    if fx["kind"] == "v1":
        return Distillation(scfg, tcfg, DistillationArgs(queue_size=fx["queue_size"], optimizer="adamw", lr=fx["lr"], weight_decay=fx["weight_decay"]), **kw)
    return DistillationV2(scfg, tcfg, DistillationV2Args(optimizer="adamw", lr=fx["lr"], weight_decay=fx["weight_decay"], n_projection_layers=fx.get("n_projection_layers", 1),
                                                          projection_hidden_dim=fx.get("projection_hidden_dim", 2048)), **kw)


@pytest.mark.parametrize("name", ["distill_v1_d64", "distill_v2_d64", "distill_v2_d64_mlp3"])
def test_distillation_v1_v2_steps_match_reference_fixture(name):
    fx = torch.load(os.path.join(GOLD, name + ".pt"), weights_only=False)
    m = build(fx)
    for si, rec in enumerate(fx["steps"]):
        x = torch.randn(fx["b"], 3, fx["img"], fx["img"], generator=torch.Generator().manual_seed(rec["x_seed"]))
        torch.manual_seed(400 + si)       # the step draws lambda and the permutation itself, in the reference's order
        res = m.training_step_impl({"views": [x]}, 0)
        assert m._last["lam"] == pytest.approx(rec["lam"]) and torch.equal(m._last["index"], rec["index"])
        assert float(res.loss) == pytest.approx(rec["logs"]["loss"], rel=1e-2)
        m.optimizer_step()
        assert float(m.last_grad_norm.sqrt()) == pytest.approx(rec["logs"]["grad_norm"], rel=5e-2)
    sd = m.state_dict()
    ref_keys = [k for k in fx["state_dict_keys"] if not k.startswith("teacher_embedding_model.")]     # on_save_checkpoint drops the teacher
    assert sorted(sd) == sorted(ref_keys)
    fin = fx["final"]
    lr = fx["steps"][-1]["logs"]["lr"]
    agree = tot = 0
    pairs = [("student_embedding_model.wrapped_model._model." + k, v, fx["init"]["student_backbone"][k]) for k, v in fin["student_backbone"].items()]
    pairs += [("student_projection_head." + k, v, fx["init"]["head"][k]) for k, v in fin["head"].items()]
    for key, v, init in pairs:
        if (v - init).abs().max().item() == 0:
            continue
        agree += int(((sd[key].cpu() - v).abs() <= 0.15 * lr * 3).sum()); tot += v.numel()
    assert agree / tot > 0.95, agree / tot
    if fx["kind"] == "v1":
        assert rel(m.teacher_queue, fin["queue"]) < 1e-2


def test_distillation_v1_with_lars_matches_reference_fixture():
    """The method's "auto" optimizer: the reference's Distillation class with DistillationLARSArgs (lr 1.8, momentum 0.9, weight decay
    1e-6) around the restated lightly.utils.lars.LARS (oracle/lars_oracle.py), 3 steps.  The first step moves the no-decay tensors by
    plain SGD at lr 0.065 on a unit-norm gradient and takes the loss from 0.81 to 1.4e-3: later losses are compared loosely, the
    three-step parameter UPDATE per tensor in the norm."""
    fx = torch.load(os.path.join(GOLD, "distill_v1_d64_lars.pt"), weights_only=False)
    assert fx["optimizer"] == "lars" and fx["lr"] == 1.8 and fx["weight_decay"] == 1e-6
    m = build(fx)
    assert m.exp_avg is None and m.lars is not None
    for si, rec in enumerate(fx["steps"]):
        x = torch.randn(fx["b"], 3, fx["img"], fx["img"], generator=torch.Generator().manual_seed(rec["x_seed"]))
        torch.manual_seed(400 + si)
        res = m.training_step_impl({"views": [x]}, 0)
        assert float(res.loss) == pytest.approx(rec["logs"]["loss"], rel=1e-2 if si == 0 else 0.25), si
        m.optimizer_step()
        assert float(m.last_grad_norm.sqrt()) == pytest.approx(rec["logs"]["grad_norm"], rel=5e-2 if si == 0 else 0.3), si
    sd = m.state_dict()
    fin = fx["final"]
    pairs = [("student_embedding_model.wrapped_model._model." + k, v, fx["init"]["student_backbone"][k]) for k, v in fin["student_backbone"].items()]
    pairs += [("student_projection_head." + k, v, fx["init"]["head"][k]) for k, v in fin["head"].items()]
    num = den = 0.0
    for key, v, init in pairs:
        upd = (v - init).double()
        if upd.abs().max().item() == 0:
            assert torch.equal(sd[key].cpu(), v), key          # tensors without gradient (v1 trains through the cls token only) stay put
            continue
        err = (sd[key].cpu().double() - v.double()).norm().item()
        num += err ** 2; den += upd.norm().item() ** 2
        assert err <= 0.12 * upd.norm().item() + 1e-7, (key, err, upd.norm().item())
    assert (num / den) ** 0.5 < 0.05


@pytest.mark.parametrize("name", ["distill_v1_d64", "distill_v2_d64"])
def test_distillation_v1_v2_gradients_match_oracle(name):
    from oracle import distill_oracle as OD

    fx = torch.load(os.path.join(GOLD, name + ".pt"), weights_only=False)
    m = build(fx)
    o = OD.OracleDistillation12(fx["kind"], fx["init"]["student_backbone"], fx["student_cfg"], fx["teacher_state"], fx["teacher_cfg"], fx["init"]["head"],
                                fx["queue_size"], fx["b"], fx["total_steps"], lr=fx["lr"], weight_decay=fx["weight_decay"])
    rec = fx["steps"][0]
    x = torch.randn(fx["b"], 3, fx["img"], fx["img"], generator=torch.Generator().manual_seed(rec["x_seed"]))
    torch.manual_seed(400)
    res = m.training_step_impl({"views": [x]}, 0)
    loss = o.forward_loss(x, rec["lam"], rec["index"])
    loss.backward()
    assert float(res.loss) == pytest.approx(float(loss.detach()), rel=1e-2)
    for n in m.student.names:
        ref = o.sb[n[9:]].grad if n.startswith("backbone.") else o.head[n[5:]].grad
        if ref is None or ref.abs().max().item() == 0:
            assert m.student.g[n].abs().max().item() == 0, n      # v1 trains on the cls token only: e.g. no gradient where none flows
            continue
        assert rel(m.student.g[n].cpu(), ref) < 5e-2, n


def test_distillation_v2_with_resnet_student_runs_and_resizes():
    """The convolutional student through v2: layer4 map (2x2 at 64^2) -> head -> bilinear resize onto the teacher's 4x4 grid -> MSE;
    loss decreases over a few steps, state_dict uses the ResNet wrapper's keys."""
    import lightly_train_amd  # noqa: F401
    from lightly_train_amd.distillation import DistillationV2, DistillationV2Args
    from lightly_train_amd.resnet import ResNetConfig
    from lightly_train_amd.vit import ViTConfig

    tcfg = ViTConfig(embed_dim=64, depth=3, num_heads=1, mlp_ratio=4.0, patch_size=16, img_size=64, init_values=0.5)
    m = DistillationV2(ResNetConfig(layers=(1, 1, 1, 1), width=8), tcfg, DistillationV2Args(optimizer="adamw", lr=0.02), global_batch_size=1536, total_steps=40, max_epochs=1,
                       device="cuda", seed=3)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(16, 3, 64, 64, generator=g)
    torch.manual_seed(0)
    losses = [float(m.train_step(x, mix=(1.0, torch.arange(16))).loss) for _ in range(12)]
    assert all(l == l for l in losses) and losses[-1] < 0.9 * losses[0], losses
    sd = m.state_dict()
    assert "student_embedding_model.wrapped_model._features.layer4.0.conv3.weight" in sd and sd["student_projection_head.mlp.weight"].shape == (128, 256)
