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


def build(fx):
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
    args = DistillationV3Args(queue_size=fx["queue_size"], weight_decay=fx["weight_decay"])
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


@pytest.mark.parametrize("arch,B,S", [("_resnet_test", 8, 64), ("resnet50", 4, 64), ("resnet50", 2, 224)])
def test_resnet_engine_matches_bf16_emulation(arch, B, S):
    """The HIP ResNet engine (im2col / MFMA GEMM / BatchNorm / max-pool kernels in context, forward and backward through every
    layer) against the SAME pipeline with plain-torch stand-ins for the ops (tests/tools/ops_emu.py: identical bf16 rounding
    points, fp32 accumulation) run on the CPU.  The orchestration itself is proven against torch autograd of the restated
    torchvision ResNet in fp32 by tests/test_resnet_engine_cpu.py; comparing bf16 against fp32 directly is meaningless for this
    network family (torch's own bf16 autocast deviates by ~40 % per gradient tensor on it)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
    import lightly_train_amd  # noqa: F401
    import ops_emu
    from lightly_train_amd import ops
    from lightly_train_amd.params import FlatParams
    from lightly_train_amd.resnet import ARCHS, ResNetConfig, ResNetEngine, flat_named
    from lightly_train_amd.vit import Workspace

    cfg = ResNetConfig(**ARCHS[arch])
    g = torch.Generator().manual_seed(5)
    sd = _perturbed_resnet_state(cfg, g)
    x = torch.randn(B, 3, S, S, generator=g)
    C = cfg.feature_dim
    outs = []
    for dev in ("cuda", "cpu"):
        import contextlib
        with (ops_emu.emulate(ops) if dev == "cpu" else contextlib.nullcontext()):
            fp = FlatParams(flat_named(cfg, sd), dev, True)
            eng = ResNetEngine(cfg, fp, "", buffers=sd)
            ws = Workspace(torch.device(dev))
            ctx = eng.forward(ws, "r", x.to(dev), save=True, train=True)
            n = B * ctx["h"] * ctx["w"]
            if dev == "cuda":
                d = torch.randn(n, C, generator=g) * 0.1
            dfeat = torch.zeros_like(ctx["feat"])
            dfeat[:n] = d.to(torch.bfloat16).to(dev)
            fp.grad.zero_()
            eng.backward(ws, ctx, dfeat)
            if dev == "cuda":
                torch.cuda.synchronize()
            outs.append((ctx["feat"][:n].float().cpu(), {k: fp.g[k].float().cpu().clone() for k in fp.names},
                         {k: v.float().cpu().clone() for k, v in eng.buffers.items()}))
    (f_hip, g_hip, b_hip), (f_emu, g_emu, b_emu) = outs
    assert rel(f_hip, f_emu) < 3e-2
    bad = {k: rel(g_hip[k], g_emu[k]) for k in g_hip}
    bad = {k: v for k, v in bad.items() if not v < 5e-2}
    assert not bad, (len(bad), sorted(bad.items(), key=lambda t: -t[1])[:6])
    for k in b_hip:
        if not k.endswith("num_batches_tracked"):
            assert rel(b_hip[k], b_emu[k]) < 1e-2, k


def test_resnet50_engine_exports_torchvision_state_and_runs_eval_mode():
    """resnet50 (3,4,6,3 bottlenecks, 25.6 M parameters): the exported state_dict has torchvision's keys, order and
    [Cout, Cin, kh, kw] layouts (parameters bit-identical, BatchNorm running statistics updated by one training forward as
    torch updates them), and the eval-mode forward (running statistics) matches the restated torchvision module."""
    import lightly_train_amd  # noqa: F401
    from lightly_train_amd.params import FlatParams
    from lightly_train_amd.resnet import ResNetConfig, ResNetEngine, flat_named, state_dict_order
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
    eng.forward(ws, "r", x.cuda(), save=True, train=True)
    ref_m = OR.resnet50()
    ref_m.load_state_dict(sd)
    ref_m.train()
    with torch.no_grad():
        OR.features(ref_m, x)
    out = eng.state_dict(extra={"fc.weight": sd["fc.weight"], "fc.bias": sd["fc.bias"]})
    assert list(out) == state_dict_order(cfg) == list(ref_m.state_dict())
    for k, v in ref_m.state_dict().items():
        assert tuple(out[k].shape) == tuple(v.shape), k
        if k.endswith("running_mean"):
            assert (out[k].cpu() - v).abs().max().item() < 3e-2 * max(1.0, v.abs().max().item()), k   # 0.1 x batch mean of bf16 activations
        elif k.endswith("running_var"):
            assert rel(out[k], v) < 5e-2, k
        elif k.endswith("num_batches_tracked"):
            assert int(out[k]) == int(v) == 1
        else:
            assert torch.equal(out[k].cpu(), sd[k]), k            # parameters come back bit-identical in torch layout
    # eval mode (running statistics, no batch coupling): well-conditioned, compared with fp32 directly
    ref_m.eval()
    eng.load_state_dict({k: v for k, v in ref_m.state_dict().items()})
    with torch.no_grad():
        fe = OR.features(ref_m, x)
    ce = eng.forward(ws, "re", x.cuda(), save=False, train=False)
    n = B * ce["h"] * ce["w"]
    assert (ce["h"], ce["w"]) == (4, 4)
    assert rel(ce["feat"][:n].float().cpu().view(B, 4, 4, 2048).permute(0, 3, 1, 2), fe) < 6e-2
