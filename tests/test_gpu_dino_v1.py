"""-m gpu: DINO (v1) in HIP (lightly_train_amd/dino.py) against tests/golden/dino_v1_d64*.pt, written by the reference's
own `DINO` class on CPU (oracle/make_golden.py::make_dino_v1: DINOv2 ViT D=64 /16, 2 x 96^2 + 2 x 48^2 views, batch 8, 4 steps, the last
layer frozen during the first two; SGD = the method's "auto" optimizer, and AdamW with a scheduled weight decay).  Tolerances (bf16 MFMA
operands vs fp32): loss 1e-2 relative, gradient norm 5e-2, first-step gradients 5e-2 of max|grad| per tensor, head outputs 2e-2 of their
range; after 4 steps the parameter UPDATE of every tensor within 12 % of its own norm (SGD) / 95 % of the elements within 0.15 lr per
step (AdamW), teacher (EMA) likewise; the center to 1e-2 of its range."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def build(fx):
    import lightly_train_amd  # noqa: F401
    from lightly_train_amd.dino import DINO, DINOArgs
    from lightly_train_amd.vit import ViTConfig

    c, ma, oa = fx["cfg"], fx["method_args"], fx["optimizer_args"]
    cfg = ViTConfig(embed_dim=c["embed_dim"], depth=c["depth"], num_heads=c["num_heads"], mlp_ratio=4.0, patch_size=c["patch_size"], img_size=c["img_size"],
                    init_values=c["init_values"])
    args = DINOArgs(hidden_dim=ma["hidden_dim"], bottleneck_dim=ma["bottleneck_dim"], output_dim=ma["output_dim"],
                    student_freeze_last_layer_steps=ma["student_freeze_last_layer_steps"], norm_last_layer=ma["norm_last_layer"],
                    teacher_temp=ma["teacher_temp"], warmup_teacher_temp=ma["warmup_teacher_temp"], warmup_teacher_temp_steps=ma["warmup_teacher_temp_steps"],
                    student_temp=ma["student_temp"], center_momentum=ma["center_momentum"], momentum_start=ma["momentum_start"], momentum_end=ma["momentum_end"],
                    weight_decay_start=ma["weight_decay_start"], weight_decay_end=ma["weight_decay_end"], warmup_steps=ma["warmup_steps"],
                    optimizer=fx["optimizer"], lr=oa["lr"], weight_decay=oa["weight_decay"])
    init = fx["init"]
    return DINO(cfg, args, global_batch_size=fx["b"], total_steps=fx["total_steps"], device="cuda", backbone_state=init["student_backbone"],
                teacher_backbone_state=init["teacher_backbone"], student_head_state=init["student_head"], teacher_head_state=init["teacher_head"])


def views_of(fx, rec):
    g = torch.Generator().manual_seed(rec["view_seed"])
    return [torch.randn(fx["b"], 3, fx["g_size"], fx["g_size"], generator=g) for _ in range(2)] + \
           [torch.randn(fx["b"], 3, fx["l_size"], fx["l_size"], generator=g) for _ in range(fx["n_local"])]


@pytest.mark.parametrize("name", ["dino_v1_d64", "dino_v1_d64_adamw"])
def test_dino_v1_steps_match_reference_fixture(name):
    from lightly_train_amd import _lib
    assert _lib.load() is not None
    fx = torch.load(os.path.join(GOLD, name + ".pt"), weights_only=False)
    m = build(fx)
    B = fx["b"]
    for si, rec in enumerate(fx["steps"]):
        res = m.training_step_impl({"views": views_of(fx, rec)}, 0)
        logs = rec["logs"]
        assert float(res.loss) == pytest.approx(logs["loss"], rel=1e-2), si
        sw = torch.cat([rec["teacher_logits"][B:], rec["teacher_logits"][:B]])
        rng = float(sw.abs().max())
        assert float((m._last["t_logits"][:2 * B].cpu() - sw).abs().max()) < 2e-2 * rng, si
        assert float((m._last["s_global_logits"].cpu() - rec["student_global_logits"]).abs().max()) < 2e-2 * rng, si
        assert float((m._last["s_local_logits"].cpu() - rec["student_local_logits"]).abs().max()) < 2e-2 * rng, si
        if "grads" in rec:
            clip = min(1.0, 3.0 / (logs["grad_norm"] + 1e-6))
            worst = 0.0
            for n in m.student.names:
                key = m._ref_key("student", n)
                if key in rec["no_grad"]:
                    assert float(m.student.g[n].abs().max()) == 0.0, n
                    continue
                want = rec["grads"][key] / clip
                err = float((m.student.g[n].cpu() - want).abs().max()) / (float(want.abs().max()) + 1e-20)
                worst = max(worst, err)
                assert err < 5e-2, (n, err)
            print(f"{name}: first-step gradients, worst tensor {worst:.3e} of max|grad|")
        m.optimizer_step()
        assert float(m.last_grad_norm.sqrt()) == pytest.approx(logs["grad_norm"], rel=5e-2), si
        c = rec["center"].view(-1)
        assert float((m.center.view(-1).cpu() - c).abs().max()) < 1e-2 * float(c.abs().max()), si
    sd = {k: v.cpu() for k, v in m.state_dict().items()}
    assert list(sd) == fx["state_dict_keys"]
    fin, init = fx["final"], fx["init"]
    n_steps = len(fx["steps"])
    lr_last = fx["steps"][-1]["hparams"]["params"]["lr"]
    for role in ("student", "teacher"):
        num = den = 0.0
        agree = tot = 0
        for part, pre in (("backbone", f"{role}_embedding_model.wrapped_model._model."), ("head", f"{role}_projection_head.")):
            for k, v in fin[f"{role}_{part}"].items():
                v0 = init[f"{role}_{part}"][k]
                upd = (v - v0).double()
                got = sd[pre + k]
                if float(upd.abs().max()) == 0:
                    assert torch.equal(got, v), (role, k)     # the mask token and the normalised last layer's weight_g never move
                    continue
                err = float((got.double() - v.double()).norm())
                num += err ** 2; den += float(upd.norm()) ** 2
                if fx["optimizer"] == "sgd" or role == "teacher":
                    # (+ fp32 round-off of the tensor itself: the EMA teacher moves by less than an ulp per element in four steps)
                    assert err <= 0.12 * float(upd.norm()) + 1e-6 * float(v.double().norm()) + 1e-7, (role, k, err, float(upd.norm()))
                else:
                    agree += int(((got - v).abs() <= 0.15 * lr_last * n_steps).sum()); tot += v.numel()
        assert (num / den) ** 0.5 < (0.06 if role == "student" else 0.12), (role, (num / den) ** 0.5)
        if tot:
            assert agree / tot > 0.95, (role, agree / tot)
        print(f"{name}: {role} update error {100 * (num / den) ** 0.5:.2f} % of the update norm")


def test_dino_v1_sgd_kernel_matches_torch_sgd():
    """lt_sgd_flat against torch.optim.SGD (momentum 0.9, coupled weight decay) over three steps, with one segment at lr 0 (its
    momentum buffer must keep moving) and clipping."""
    from lightly_train_amd import ops
    from lightly_train_amd.params import FlatParams

    g = torch.Generator().manual_seed(5)
    named = [("a", torch.randn(300, 7, generator=g)), ("b", torch.randn(2048, generator=g)), ("c", torch.randn(33, generator=g))]
    fp = FlatParams(named, "cuda", True)
    ref = [torch.nn.Parameter(t.clone().cuda()) for _, t in named]
    opt = torch.optim.SGD([{"params": [ref[0]], "weight_decay": 1e-2}, {"params": [ref[1]], "weight_decay": 0.0}, {"params": [ref[2]], "weight_decay": 1e-2, "lr": 0.0}],
                          lr=0.1, momentum=0.9)
    seg_lr = torch.tensor([0.1, 0.1, 0.0], device="cuda")
    seg_wd = torch.tensor([1, 0, 1], dtype=torch.uint8, device="cuda")
    buf = torch.zeros_like(fp.data)
    sumsq = torch.zeros(1, device="cuda")
    for step in range(3):
        fp.grad.zero_()
        for (n, _), r in zip(named, ref):
            gr = torch.randn(r.shape, generator=g).cuda()
            fp.g[n].copy_(gr)
            r.grad = gr.clone()
        torch.nn.utils.clip_grad_norm_(ref, 3.0)
        opt.step()
        sumsq.zero_()
        ops.sumsq(fp.grad, sumsq)
        ops.sgd_flat(fp.data, fp.grad, buf, fp.bf16, fp.seg_of_chunk, seg_lr, seg_wd, 1.0, 1e-2, 0.9, 0.0, False, step == 0, sumsq, 3.0)
    for (n, _), r in zip(named, ref):
        assert torch.allclose(fp.p[n], r.data, atol=2e-6), n
        o, cnt = fp.offsets[n], r.numel()
        assert torch.allclose(buf[o:o + cnt].view(r.shape), opt.state[r]["momentum_buffer"], atol=2e-6), n
        assert torch.equal(fp.b[n].float(), fp.p[n].to(torch.bfloat16).float()), n
    assert torch.equal(fp.p["c"].cpu(), named[2][1])     # lr 0: parameters untouched


def test_dino_v1_on_a_resnet_steps_match_reference_fixture():
    """`DINOResNet` in HIP against tests/golden/dino_v1_resnet.pt (the reference's DINO class around its ResNetModelWrapper; width-8
    bottleneck ResNet, 96^2 / 48^2 views, batch 8).  The conv path at this size is the noisy regime DESIGN 3 measures (BatchNorm over
    16 x 3 x 3 positions in layer4 amplifies bf16 rounding): the fixture carries the reference's OWN first step under torch.autocast(bf16)
    against its fp32 step -- per-tensor gradient error median 0.30, max 0.62 -- and the HIP step is held to that yardstick (median <= 1.25 x,
    max <= 1.5 x), next to the loss per step (3e-2), the gradient norm (10 %), the head outputs (5 % of their range) and the 4-step update
    in the norm; exact arithmetic is checked on the CPU (tests/test_dino_v1_cpu.py)."""
    import json

    from lightly_train_amd.dino import DINOArgs, DINOResNet
    from lightly_train_amd.resnet import ResNetConfig, from_flat_layout

    fx = torch.load(os.path.join(GOLD, "dino_v1_resnet.pt"), weights_only=False)
    ma, oa, c, init = fx["method_args"], fx["optimizer_args"], fx["cfg"], fx["init"]
    args = DINOArgs(hidden_dim=ma["hidden_dim"], bottleneck_dim=ma["bottleneck_dim"], output_dim=ma["output_dim"],
                    student_freeze_last_layer_steps=ma["student_freeze_last_layer_steps"], teacher_temp=ma["teacher_temp"],
                    warmup_teacher_temp=ma["warmup_teacher_temp"], warmup_teacher_temp_steps=ma["warmup_teacher_temp_steps"], momentum_start=ma["momentum_start"],
                    weight_decay_start=ma["weight_decay_start"], weight_decay_end=ma["weight_decay_end"], optimizer=fx["optimizer"], lr=oa["lr"],
                    weight_decay=oa["weight_decay"])
    m = DINOResNet(ResNetConfig(layers=tuple(c["layers"]), width=c["width"]), args, global_batch_size=fx["b"], total_steps=fx["total_steps"], device="cuda",
                   backbone_state=init["student_backbone"], teacher_backbone_state=init["teacher_backbone"], student_head_state=init["student_head"],
                   teacher_head_state=init["teacher_head"])
    B = fx["b"]
    report = {"steps": []}
    for si, rec in enumerate(fx["steps"]):
        res = m.training_step_impl({"views": views_of(fx, rec)}, 0)
        logs = rec["logs"]
        rng = float(rec["teacher_logits"].abs().max())
        e_t = float((m._last["t_logits"][:2 * B].cpu() - rec["teacher_logits"]).abs().max()) / rng
        e_s = float((m._last["s_global_logits"].cpu() - rec["student_global_logits"]).abs().max()) / rng
        entry = {"loss": float(res.loss), "ref_loss": logs["loss"], "teacher_logits_err": e_t, "student_logits_err": e_s}
        assert float(res.loss) == pytest.approx(logs["loss"], rel=3e-2), si
        assert e_t < 5e-2 and e_s < 5e-2, (si, e_t, e_s)
        if "grads" in rec:
            clip = min(1.0, 3.0 / (logs["grad_norm"] + 1e-6))
            errs = []
            for n in m.student.names:
                key = m._ref_key("student", n)
                if key in rec["no_grad"]:
                    continue
                want = rec["grads"][key] / clip
                got = (from_flat_layout(n[9:], m.student.g[n]) if n.startswith("backbone.") else m.student.g[n]).cpu()
                errs.append(float((got - want).abs().max()) / (float(want.abs().max()) + 1e-20))
            errs.sort()
            yard = sorted(rec["bf16_autocast"]["grad_err"].values())      # the reference's own bf16-autocast step against its fp32 step
            entry["grad_err_median"], entry["grad_err_max"] = errs[len(errs) // 2], errs[-1]
            entry["reference_autocast_grad_err_median"], entry["reference_autocast_grad_err_max"] = yard[len(yard) // 2], yard[-1]
            assert errs[len(errs) // 2] <= 1.25 * yard[len(yard) // 2] + 0.02 and errs[-1] <= 1.5 * yard[-1] + 0.05, (entry)
        m.optimizer_step()
        entry["grad_norm"], entry["ref_grad_norm"] = float(m.last_grad_norm.sqrt()), logs["grad_norm"]
        assert entry["grad_norm"] == pytest.approx(logs["grad_norm"], rel=0.10), si
        report["steps"].append(entry)
    sd = {k: v.cpu() for k, v in m.state_dict().items()}
    assert list(sd) == fx["state_dict_keys"]
    num = den = 0.0
    worst = 0.0
    for k, v in fx["final"]["student_backbone"].items():
        v0 = init["student_backbone"][k]
        if not v.is_floating_point() or float((v - v0).abs().max()) == 0:
            continue
        got = sd["student_embedding_model.wrapped_model._features." + k]
        upd = float((v - v0).double().norm())
        err = float((got.double() - v.double()).norm())
        num += err ** 2; den += upd ** 2
        if "running" not in k:
            worst = max(worst, err / upd)
    report["update_err_total"], report["update_err_worst_tensor"] = (num / den) ** 0.5, worst
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(report, open(os.path.join(ROOT, "gpurun_out", "dino_v1_resnet_report.json"), "w"), indent=1)
    print(json.dumps(report))
    # the whole-model update (dominated by the BatchNorm running estimates) to 1 %; a single tensor's update no worse than the yardstick's
    # per-tensor gradient error allows (observed: 0.2 % in total, 0.48 for the worst 4-step update against 0.62 for the worst autocast gradient)
    yard_max = max(fx["steps"][0]["bf16_autocast"]["grad_err"].values())
    assert report["update_err_total"] < 0.01 and worst <= 1.25 * yard_max, report
