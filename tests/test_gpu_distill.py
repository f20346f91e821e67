"""-m gpu: the DistillationV3 step (frozen DINOv3 ViT teacher with RoPE -> DINOv2-ViT student) in HIP
against tests/golden/distill_v3_d64.pt, written by the reference's own DistillationV3 class on CPU (oracle/make_golden.py),
and against the oracle restatement for gradients.

Tolerances (bf16 MFMA operands vs fp32): global KL 1e-2 relative; local KL (a ~1e-3 quantity made of 14 x 14 similarity
softmaxes at temperature 0.07) 25 % relative / 3e-4 absolute; gradient norm 5e-2; gradients 5e-2 of max|grad| per tensor;
teacher queue 1e-2; after 3 AdamW steps > 95 % of the parameter updates within 0.15 lr of the reference's."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-20)).item()


def build(fx, **args_over):
    import lightly_train_amd  # noqa: F401
    from lightly_train_amd.dinov3 import convert_dinov3_state, dinov3_vit_config
    from lightly_train_amd.distillationv3 import DistillationV3, DistillationV3Args
    from lightly_train_amd.vit import ViTConfig

    sc, tc = fx["student_cfg"], fx["teacher_cfg"]
    student_state = fx["init"]["student_backbone"]
    if sc.get("kind") == "resnet":      # convolutional student (torchvision ResNet keys / layouts)
        from lightly_train_amd.resnet import ResNetConfig

        scfg = ResNetConfig(layers=tuple(sc["layers"]), width=sc["width"])
    elif sc.get("kind") == "dinov3":      # DINOv3 student: RoPE with the training-mode rescale, storage tokens, K-masked bias
        scfg = dinov3_vit_config(sc["embed_dim"], sc["depth"], sc["num_heads"], patch_size=sc["patch_size"], img_size=sc["img_size"],
                                 n_storage_tokens=sc["n_storage_tokens"], layerscale_init=sc["init_values"], rope_base=sc["rope_base"],
                                 ln_eps=sc["ln_eps"], rope_rescale=sc["rope_rescale"])
        student_state = convert_dinov3_state(student_state, scfg)
    else:
        scfg = ViTConfig(embed_dim=sc["embed_dim"], depth=sc["depth"], num_heads=sc["num_heads"], mlp_ratio=4.0, patch_size=sc["patch_size"],
                         img_size=sc["img_size"], init_values=sc["init_values"])
    tcfg = dinov3_vit_config(tc["embed_dim"], tc["depth"], tc["num_heads"], patch_size=tc["patch_size"], img_size=tc["img_size"],
                             n_storage_tokens=tc["n_storage_tokens"], layerscale_init=0.5, rope_base=tc["rope_base"], ln_eps=tc["ln_eps"])
    args = DistillationV3Args(queue_size=fx["queue_size"], weight_decay=fx["weight_decay"], **args_over)
    return DistillationV3(scfg, tcfg, args, global_batch_size=fx["b"], total_steps=fx["total_steps"], max_epochs=1, device="cuda",
                          student_state=student_state, teacher_state=convert_dinov3_state(fx["teacher_state"], tcfg),
                          proj_global_state=fx["init"]["proj_global"], proj_local_state=fx["init"]["proj_local"])


# equal grids / 8x8 student grid resized onto 7x7 / DINOv3 student (training-mode RoPE rescale draws) / ResNet student (2x2 map -> 4x4)
@pytest.mark.parametrize("name", ["distill_v3_d64", "distill_v3_d64_p14", "distill_v3_d64_v3s", "distill_v3_resnet"])
def test_distillation_step_matches_reference_fixture(name):
    fx = torch.load(os.path.join(GOLD, name + ".pt"), weights_only=False)
    m = build(fx)
    img = fx.get("img", 64)
    for si, rec in enumerate(fx["steps"]):
        x = torch.randn(fx["b"], 3, img, img, generator=torch.Generator().manual_seed(rec["x_seed"]))
        torch.manual_seed(300 + si)     # the generator's seed: the step draws lambda, the permutation (and RoPE rescales) itself
        res = m.training_step_impl({"views": [x]}, 0)
        assert m._last["lam"] == pytest.approx(rec["lam"]) and torch.equal(m._last["index"], rec["index"])
        logs = {k.split("/")[-1]: float(v) for k, v in res.log_dict.items()}
        assert logs["global_loss"] == pytest.approx(rec["logs"]["global_loss"], rel=1e-2)
        assert logs["local_loss"] == pytest.approx(rec["logs"]["local_loss"], rel=0.25, abs=3e-4)
        m.optimizer_step()
        assert float(m.last_grad_norm.sqrt()) == pytest.approx(rec["logs"]["grad_norm"], rel=5e-2)
    fin = fx["final"]
    assert rel(m.teacher_queue, fin["queue"]) < 1e-2
    sd = m.state_dict()
    agree = tot = 0
    lr = fx["steps"][-1]["logs"]["lr"]
    conv = fx["student_cfg"].get("kind") == "resnet"
    key = "student_embedding_model.wrapped_model." + ("_features." if conv else "_model.")
    for k, v in fin["student_backbone"].items():
        if conv and k.startswith("fc."):
            continue                     # the classifier is not part of the wrapper's trained / saved modules
        ours = sd[key + k].cpu()
        if k.endswith("num_batches_tracked"):
            assert int(ours) == int(v) == len(fx["steps"])
            continue
        if k.endswith(("running_mean", "running_var")):    # BatchNorm running statistics after 3 training forwards
            assert rel(ours, v) < 2e-2, k
            continue
        init = fx["init"]["student_backbone"][k]
        if (v - init).abs().max().item() == 0:
            continue
        agree += int(((ours - v).abs() <= 0.15 * lr * 3).sum()); tot += v.numel()
    if conv:
        # BatchNorm networks at random init amplify bf16 rounding: the fixture records that the reference's OWN bf16-mixed path
        # (CPU autocast, oracle/make_golden.py::_reference_bf16_resnet_run) agrees with its fp32 run on only 79 % of the updates by
        # this very measure -- the HIP step has to do at least as well as that (observed 80 %)
        yard = fx["reference_bf16"]["update_agreement"]
        assert 0.7 < yard < 0.9
        assert agree / tot > yard - 0.03, (agree / tot, yard)
    else:
        assert agree / tot > 0.95, agree / tot
    assert "student_projection_head_local.weight" in sd and "teacher_queue" in sd


@pytest.mark.parametrize("name", ["distill_v3_d64", "distill_v3_resnet"])
def test_distillationv3_lars_option_steps_like_the_flat_rule_on_the_same_gradients(name):
    """optimizer="lars" (DistillationV3LARSArgs, distillationv3.py:147-157): the method hands its clipped gradients, lr schedule and
    decay groups to lars.FlatLARS (the rule itself: test_gpu_ops.py::test_lars_flat_matches_oracle; the reference's orchestration of it:
    the Distillation v1 fixture).  Here: the step taken equals the rule applied by hand to the gradients the method produced."""
    from oracle.lars_oracle import LARS

    fx = torch.load(os.path.join(GOLD, name + ".pt"), weights_only=False)
    m = build(fx, optimizer="lars")
    assert m.lars is not None and m.exp_avg is None and m.base_lr == pytest.approx(1.8 * (fx["b"] / 1536) ** 0.5)
    rec = fx["steps"][0]
    x = torch.randn(fx["b"], 3, fx.get("img", 64), fx.get("img", 64), generator=torch.Generator().manual_seed(rec["x_seed"]))
    torch.manual_seed(300)
    losses = []
    for step in range(3):
        res = m.training_step_impl({"views": [x]}, 0)
        losses.append(float(res.loss))
        before = {n: m.student.p[n].detach().cpu().clone().requires_grad_(True) for n in m.student.names}
        for n in m.student.names:
            before[n].grad = m.student.g[n].detach().cpu().clone()
        if step == 0:
            dec = [n for i, n in enumerate(m.student.names) if int(m.seg_wd_on[i])]
            nod = [n for i, n in enumerate(m.student.names) if not int(m.seg_wd_on[i])]
            ref_params = before
            ref = LARS([{"params": [before[n] for n in dec]}, {"params": [before[n] for n in nod], "weight_decay": 0.0}], lr=1.0, momentum=0.9,
                       weight_decay=1e-6, trust_coefficient=0.001, eps=1e-8)
        else:
            for n in m.student.names:
                ref_params[n].data.copy_(before[n].data); ref_params[n].grad = before[n].grad
        from lightly_train_amd.schedules import warmup_cosine_lr_factor
        f = warmup_cosine_lr_factor(m.trainer.global_step, m.warmup_steps, int(m.trainer.estimated_stepping_batches), 0.001)
        torch.nn.utils.clip_grad_norm_(list(ref_params.values()), 1.0)
        for g_ in ref.param_groups:
            g_["lr"] = m.base_lr * f
        ref.step()
        m.optimizer_step()
        for n in m.student.names:
            assert torch.allclose(m.student.p[n].cpu(), ref_params[n].detach(), rtol=1e-4, atol=1e-6), (step, n)
    assert all(l == l for l in losses)


@pytest.mark.parametrize("name", ["distill_v3_d64", "distill_v3_d64_p14", "distill_v3_d64_v3s", "distill_v3_resnet"])
def test_distillation_gradients_match_oracle(name):
    from oracle import distill_oracle as OD

    fx = torch.load(os.path.join(GOLD, name + ".pt"), weights_only=False)
    img = fx.get("img", 64)
    m = build(fx)
    o = OD.OracleDistillationV3(fx["init"]["student_backbone"], fx["student_cfg"], fx["teacher_state"], fx["teacher_cfg"],
                                fx["init"]["proj_global"], fx["init"]["proj_local"], fx["queue_size"], fx["b"], fx["total_steps"],
                                weight_decay=fx["weight_decay"])
    rec = fx["steps"][0]
    x = torch.randn(fx["b"], 3, img, img, generator=torch.Generator().manual_seed(rec["x_seed"]))
    torch.manual_seed(300)
    res = m.training_step_impl({"views": [x]}, 0)
    loss, _ = o.forward_loss(x, rec["lam"], rec["index"], rec.get("rescales"))
    loss.backward()
    assert float(res.loss) == pytest.approx(float(loss.detach()), rel=1e-2)
    L = m._last
    with torch.no_grad():
        t = OD.O3.dinov3_vit_forward(o.teacher, rec["lam"] * x + (1 - rec["lam"]) * x[rec["index"]], o.tcfg)
        ref_tl = torch.nn.functional.normalize(t["x_norm_patchtokens"], dim=-1).flatten(0, 1)
    assert rel(L["tl"][: ref_tl.shape[0]], ref_tl) < 2e-2
    ren = {"register_tokens": "storage_tokens"} if fx["student_cfg"].get("kind") == "dinov3" else {}
    for n in m.student.names:
        if n == "backbone.pos_embed" and ren:
            assert m.student.g[n].abs().max().item() == 0      # RoPE model: the (zero) positional table is frozen
            continue
        if n.startswith("backbone.") and o.resnet is not None:
            # the convolutional backbone's gradients are ill-conditioned under bf16 (torch's own bf16 autocast of this very network
            # deviates from fp32 by ~40 % per tensor, see the fixture's reference_bf16 record): the HIP engine is checked against
            # the same arithmetic with identical rounding points instead (test_resnet_engine_matches_bf16_emulation below)
            continue
        elif n.startswith("backbone."):
            ref = o.sb[ren.get(n[9:], n[9:])].grad
        elif n.startswith("proj_global."):
            ref = o.pg[n[12:]].grad
        else:
            ref = o.pl[n[11:]].grad
        if ref is None or ref.abs().max().item() == 0:
            continue
        assert rel(m.student.g[n].cpu(), ref) < (1e-1 if o.resnet is not None else 5e-2), n


def _perturbed_resnet_state(cfg, g):
    from lightly_train_amd.resnet import init_resnet_state

    sd = init_resnet_state(cfg, g)
    for k in sd:     # BatchNorm affine away from (1, 0) so that every term of the backward is exercised
        if (".bn" in k or k.startswith("bn") or "downsample.1" in k) and k.endswith(("weight", "bias")):
            sd[k] = sd[k] + 0.2 * torch.randn(sd[k].shape, generator=g)
    return sd


def fro(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-20)).item()


@pytest.mark.parametrize("layers,width,B,S,g_med,g_max", [((1, 1, 1), 16, 16, 64, 0.10, 0.30), ((2, 2), 32, 8, 64, 0.30, 0.50),
                                                          ((3,), 64, 4, 64, 0.15, 0.35)])
def test_resnet_engine_end_to_end_against_bf16_emulation(layers, width, B, S, g_med, g_max):
    """The HIP ResNet engine end to end (stem, max-pool, bottlenecks with identity / strided downsample paths; im2col, MFMA GEMMs
    incl. the 256-row kernel at M >= 2048, BatchNorm, col2im in context; forward and backward) against the same pipeline with
    plain-torch stand-ins for the ops (tests/tools/ops_emu.py, bf16 storage at the same points) run on the CPU.

    What this comparison can and cannot show: two bf16 pipelines whose roundings differ in a few last bits (fmaf vs mul+add in
    BatchNorm, GEMM summation order) decorrelate within two or three BatchNorm layers -- every perturbation shifts the batch
    statistics, hence every element, hence thousands of bf16 roundings -- and from then on differ from each other by the bf16
    noise level itself, exactly as either differs from fp32 (torch's own bf16 autocast of these nets deviates from its fp32 run by
    ~40 % per gradient tensor; emulation vs emulation with BatchNorm evaluated in fp64 instead of fp32: feature map 0.3-1.3 %,
    gradients 3-17 % in Frobenius norm).  So the bounds here are the bf16 noise level of a random upstream gradient pushed through
    BatchNorm backward -- tight enough to expose any wrong tap order, layout, stride or reduction (those give ~100 %).  The tight
    checks are per op against torch (tests/test_gpu_ops.py: <= 1e-2) and of the orchestration in exact arithmetic against torch
    autograd of the restated torchvision module (tests/test_resnet_engine_cpu.py: 1e-3)."""
    import contextlib
    import statistics
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
    import lightly_train_amd  # noqa: F401
    import ops_emu
    from lightly_train_amd import ops
    from lightly_train_amd.params import FlatParams
    from lightly_train_amd.resnet import ResNetConfig, ResNetEngine, flat_named
    from lightly_train_amd.vit import Workspace

    cfg = ResNetConfig(layers=layers, width=width)
    g = torch.Generator().manual_seed(5)
    sd = _perturbed_resnet_state(cfg, g)
    x = torch.randn(B, 3, S, S, generator=g)
    C = cfg.feature_dim
    outs, d = [], None
    for dev in ("cuda", "cpu"):
        with (ops_emu.emulate(ops) if dev == "cpu" else contextlib.nullcontext()):
            fp = FlatParams(flat_named(cfg, sd), dev, True)
            eng = ResNetEngine(cfg, fp, "", buffers=sd)
            ws = Workspace(torch.device(dev))
            ctx = eng.forward(ws, "r", x.to(dev), save=True, train=True)
            n = B * ctx["h"] * ctx["w"]
            if d is None:
                d = torch.randn(n, C, generator=g) * 0.1
            dfeat = torch.zeros_like(ctx["feat"])
            dfeat[:n] = d.to(torch.bfloat16).to(dev)
            fp.grad.zero_()
            eng.backward(ws, ctx, dfeat)
            if dev == "cuda":
                torch.cuda.synchronize()
            outs.append((ctx["feat"][:n].float().cpu(), {k: fp.g[k].float().cpu().clone() for k in fp.names},
                         {k: v.float().cpu().clone() for k, v in eng.buffers.items()}, ctx["stem"]["c"][: ctx["stem"]["r1"]].float().cpu()))
    (f_hip, g_hip, b_hip, c_hip), (f_emu, g_emu, b_emu, c_emu) = outs
    assert fro(c_hip, c_emu) < 2e-3                      # first convolution: identical inputs on both sides, rounding flips only
    assert fro(f_hip, f_emu) < 5e-2
    errs = {k: fro(g_hip[k], g_emu[k]) for k in g_hip}
    assert statistics.median(errs.values()) < g_med and max(errs.values()) < g_max, sorted(errs.items(), key=lambda t: -t[1])[:6]
    for k in b_hip:
        if k.endswith("running_mean"):
            assert (b_hip[k] - b_emu[k]).abs().max().item() < 2e-2, k
        elif k.endswith("running_var"):
            assert fro(b_hip[k], b_emu[k]) < 3e-2, k


def test_resnet50_engine_exports_torchvision_state_and_runs():
    """resnet50 (3,4,6,3 bottlenecks, 25.6 M parameters) at 128^2, batch 8: a training forward + backward runs through all 53
    convolutions / BatchNorms with finite results and non-zero gradients everywhere; the exported state_dict has torchvision's
    keys, order and [Cout, Cin, kh, kw] layouts with bit-identical parameters and running statistics that moved; eval mode runs."""
    import lightly_train_amd  # noqa: F401
    from lightly_train_amd.params import FlatParams
    from lightly_train_amd.resnet import ResNetConfig, ResNetEngine, flat_named, resnet_param_shapes, state_dict_order
    from lightly_train_amd.vit import Workspace
    from oracle import resnet_oracle as OR

    cfg = ResNetConfig()
    g = torch.Generator().manual_seed(5)
    sd = _perturbed_resnet_state(cfg, g)
    fp = FlatParams(flat_named(cfg, sd), "cuda", True)
    eng = ResNetEngine(cfg, fp, "", buffers=sd)
    ws = Workspace(torch.device("cuda"))
    B = 8
    x = torch.randn(B, 3, 128, 128, generator=g)
    ctx = eng.forward(ws, "r", x.cuda(), save=True, train=True)
    n = B * ctx["h"] * ctx["w"]
    assert (ctx["h"], ctx["w"]) == (4, 4) and torch.isfinite(ctx["feat"][:n].float()).all()
    dfeat = torch.zeros_like(ctx["feat"])
    dfeat[:n] = (torch.randn(n, 2048, generator=g) * 0.1).to(torch.bfloat16).cuda()
    fp.grad.zero_()
    eng.backward(ws, ctx, dfeat, side=torch.cuda.Stream())
    torch.cuda.synchronize()
    for nm, _ in resnet_param_shapes(cfg):
        gr = fp.g[nm]
        assert torch.isfinite(gr).all() and float(gr.abs().max()) > 0, nm
    ref_m = OR.resnet50()
    out = eng.state_dict(extra={"fc.weight": sd["fc.weight"], "fc.bias": sd["fc.bias"]})
    assert list(out) == state_dict_order(cfg) == list(ref_m.state_dict())
    for k, v in ref_m.state_dict().items():
        assert tuple(out[k].shape) == tuple(v.shape), k
        if k.endswith(("running_mean", "running_var")):
            assert not torch.equal(out[k].cpu(), sd[k]), k       # updated by the training forward
        elif k.endswith("num_batches_tracked"):
            assert int(out[k]) == 1
        else:
            assert torch.equal(out[k].cpu(), sd[k]), k            # parameters come back bit-identical in torch layout
    ref_m.load_state_dict(out)                                    # strict: a torchvision-compatible export
    ce = eng.forward(ws, "re", x.cuda(), save=False, train=False)
    assert torch.isfinite(ce["feat"][:n].float()).all()


@pytest.mark.parametrize("name", ["distill_v3_d64", "distill_v3_resnet"])
def test_distillationv3_state_round_trip_and_resume(name):
    """f4 for DistillationV3: state_dict() + optimizer_state() after one step load into an object built from another random state; the next
    step of both agrees to the last bits (the CPU test in exact arithmetic asks for bit equality: tests/test_distillation_methods_cpu.py)."""
    fx = torch.load(os.path.join(GOLD, name + ".pt"), weights_only=False)
    img = fx.get("img", 64)

    def step(m, si):
        x = torch.randn(fx["b"], 3, img, img, generator=torch.Generator().manual_seed(fx["steps"][si]["x_seed"]))
        torch.manual_seed(300 + si)
        res = m.training_step_impl({"views": [x]}, 0)
        m.optimizer_step()
        return float(res.loss)

    a = build(fx)
    step(a, 0)
    sd, ost = a.state_dict(), a.optimizer_state()
    fx2 = dict(fx, init=dict(fx["init"], student_backbone={k: (v + 0.01 * torch.randn_like(v) if v.is_floating_point() else v)
                                                             for k, v in fx["init"]["student_backbone"].items()}))
    b = build(fx2)
    b.load_state_dict(sd)
    b.load_optimizer_state(ost)
    for k, v in sd.items():
        assert torch.equal(b.state_dict()[k].cpu(), v.cpu()), k
    la, lb = step(a, 1), step(b, 1)
    assert la == pytest.approx(lb, rel=1e-6)
    fa, fb = a.state_dict(), b.state_dict()
    for k in fa:   # the small weight-gradient GEMMs reduce their K-slices with fp32 atomics: last-bit differences between two runs
        assert torch.allclose(fa[k].float(), fb[k].float(), atol=2e-6, rtol=1e-5), k
