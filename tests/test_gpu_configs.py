"""-m gpu: step-level parity at the shapes the OTHER BASELINE.json configurations dispatch (round-2 verdict, "weak" item 1).

test_gpu_step.py pins the headline configuration (configs[2], ViT-B/16: `test_vitb_batch24_step_dispatches_gemm256q_and_matches_oracle`).
This file does the same for
  configs[1]  ViT-S/16 at >= 2048 token rows (batch 16): every token GEMM on `gemm256q`, ALL tensors asserted;
  configs[4]  ViT-L/14 SwiGLU at 518^2 (1370 / 1374 tokens), depth cut to 4: the partial-K-tile `gemm256q` variant (K = 2736 / 5472 are
              not multiples of 64) and the long-sequence attention kernels inside a checked step;
  configs[3]  frozen DINOv3 ViT-L/16 teacher (depth cut) -> the FULL torchvision resnet50 student at 224^2, batch 32.
The fp32 oracle restatements (oracle/*.py, pinned on the reference in tests/test_oracle_pin.py) are plain torch; for these sizes they
run on the GPU in fp32 (rocBLAS fp32 has no reduced-precision mode on gfx950) so that the comparisons finish in seconds -- the oracle
is the checker here, never the thing measured.

Tolerances (bf16 MFMA operands / fp32 accumulate vs fp32): loss terms 2e-3 relative (ViT), per-tensor gradients 4e-2 of max|grad|
(observed <= 1.8e-2 on all 183 / 71 / 72 tensors)
(KoLeo off, as everywhere at LayerScale 1e-5: DESIGN 3), gradient norm 2e-2.  ResNet-50: stated at the test."""
import json
import os
import random

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-20)).item()


def fro(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-20)).item()


def _oracle_to_gpu(o):
    """Move an OracleDINOv2's tensors to the GPU (fp32 torch ops there; leaves stay leaves)."""
    def mv(d, grad):
        for k in list(d):
            d[k] = d[k].detach().cuda().requires_grad_(grad and d[k].is_floating_point())
    shared = o.shi is o.sh
    shared_t = o.thi is o.th
    mv(o.sb, True); mv(o.sh, True); mv(o.tb, False); mv(o.th, False)
    if shared:
        o.shi = o.sh
    else:
        mv(o.shi, True)
    if shared_t:
        o.thi = o.th
    else:
        mv(o.thi, False)
    o.dino_center = o.dino_center.cuda(); o.ibot_center = o.ibot_center.cuda()
    return o


def _install_spies():
    """Record which MFMA GEMM kernel family each lt_gemm_bf16 call dispatches to (mirror of the dispatcher's size gate, gemm.hip `big` /
    `ktail`) and the sequence lengths the attention kernels are called with."""
    from lightly_train_amd import ops

    gemms, attn = [], []
    og, of, ob = ops.gemm, ops.attention_fwd, ops.attention_bwd

    def gemm(a, b, out, *, M, N, K, trans_a=False, trans_b=False, **kw):
        epi = kw.get("epilogue", 0)
        ktail = K % 64 != 0 and K % 8 == 0 and K > 64 and not trans_a and epi != ops.EPI_F32_ACCUM
        big = (K % 64 == 0 or ktail) and N % 8 == 0 and N >= (256 if ktail else 128) and (
            (not trans_a and M >= 2048) or (trans_a and K >= 4096 and M >= 64))
        kind = "wgrad" if trans_a else ("dgrad" if trans_b else "fwd")
        gemms.append((kind, ("gemm256q_ktail" if ktail else "gemm256") if big else "gemm128", M, N, K, epi))
        return og(a, b, out, M=M, N=N, K=K, trans_a=trans_a, trans_b=trans_b, **kw)

    def afwd(qkv, out, lse, B, N, H, dh, scale):
        attn.append(("fwd", N, H))
        return of(qkv, out, lse, B, N, H, dh, scale)

    def abwd(qkv, out, dout, lse, ws, dqkv, B, N, H, dh, *a, **kw):
        attn.append(("bwd", N, H))
        return ob(qkv, out, dout, lse, ws, dqkv, B, N, H, dh, *a, **kw)

    ops.gemm, ops.attention_fwd, ops.attention_bwd = gemm, afwd, abwd

    def undo():
        ops.gemm, ops.attention_fwd, ops.attention_bwd = og, of, ob

    return gemms, attn, undo


def _check_all_gradients(m, o, tol, report_name, extra=None):
    sq_o = sq_r = 0.0
    report, bad = {}, []
    for n in m.student.names:
        ref = (o.sb[n[9:]] if n.startswith("backbone.") else o.sh[n[5:]]).grad
        ours = m.student.g[n]
        sq_o += float((ours.double() ** 2).sum()); sq_r += float((ref.double() ** 2).sum())
        report[n] = rel(ours, ref)
        if not report[n] < tol:
            bad.append((n, report[n]))
    out_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out_dir):
        vals = sorted(report.values())
        with open(os.path.join(out_dir, report_name), "w") as f:
            json.dump({"max": vals[-1], "median": vals[len(vals) // 2], "n_tensors": len(vals), "grad_norm_ours": sq_o ** 0.5,
                       "grad_norm_oracle": sq_r ** 0.5, "per_tensor": report, **(extra or {})}, f, indent=1)
    assert not bad, f"{len(bad)} of {len(report)} tensors off: {sorted(bad, key=lambda t: -t[1])[:8]}"
    assert sq_o ** 0.5 == pytest.approx(sq_r ** 0.5, rel=2e-2)


def test_cfg2_vit_small_batch16_step_on_gemm256q_all_tensors():
    """BASELINE configs[1]: ViT-S/16 (D=384, 6 heads, 12 blocks, LayerScale 1e-5), 2 x 224^2 + 8 x 98^2 crops, K = 65 536, batch 16
    => 6304 global / 6400 local token rows: every forward / dgrad token GEMM takes the 256-row four-phase kernel (N = 384 = 1.5 column
    tiles, N = 1152, 1536), the token weight gradients its slab split-K form -- the kernels `bench.py --model vit_small` runs.
    Every one of the 175 parameter tensors is compared with the fp32 oracle."""
    import lightly_train_amd  # noqa: F401
    from lightly_train_amd import ops
    from lightly_train_amd.dinov2 import DINOv2, DINOv2Args, init_head_state
    from lightly_train_amd.vit import ViTConfig, init_vit_state
    from oracle import dinov2_oracle as O

    g = torch.Generator().manual_seed(52)
    vc = ViTConfig(embed_dim=384, depth=12, num_heads=6, mlp_ratio=4.0, patch_size=16, img_size=224, init_values=1e-5)
    bsd = init_vit_state(vc, g)
    shs, ths = init_head_state(384, 2048, 256, 65536, g), init_head_state(384, 2048, 256, 65536, g)
    b = 16
    m = DINOv2(vc, DINOv2Args(koleo_loss_weight=0.0), global_batch_size=b, total_steps=100, device="cuda", backbone_state=bsd,
               student_head_state=shs, teacher_head_state=ths)
    o = _oracle_to_gpu(O.OracleDINOv2(bsd, shs, dict(patch_size=16, num_heads=6, depth=12), args=dict(koleo_loss_weight=0.0),
                                      global_batch_size=b, total_steps=100, teacher_head=ths))
    views = [torch.randn(b, 3, 224, 224, generator=g) for _ in range(2)] + [torch.randn(b, 3, 98, 98, generator=g) for _ in range(8)]
    random.seed(8)
    gemms, attn, undo = _install_spies()
    try:
        res = m.training_step_impl({"views": views}, 0)
    finally:
        undo()
    torch.cuda.synchronize()
    tok = [c for c in gemms if c[0] in ("fwd", "dgrad") and c[2] in (2 * b * 197, 8 * b * 50)]
    assert len(tok) >= 12 * 4 * 3 and all(c[1] == "gemm256" for c in tok), [c for c in tok if c[1] != "gemm256"][:3]
    big = [c for c in gemms if c[1] == "gemm256"]
    assert {c[0] for c in big} == {"fwd", "dgrad", "wgrad"}
    assert {ops.EPI_BF16_GELU, ops.EPI_RESID, ops.EPI_BF16_GELUGRAD, ops.EPI_BF16, ops.EPI_F32_ACCUM} <= {c[5] for c in big}
    assert {c[3] for c in tok} >= {384, 1152, 1536}

    masks = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in m._last_masks.items()}
    loss, ologs = o.forward_loss([v.cuda() for v in views], masks)
    loss.backward()
    logs = {k.split("/")[-1]: float(v) for k, v in res.log_dict.items()}
    for k in ("dino_global_loss", "dino_local_loss", "ibot_loss"):
        assert logs[k] == pytest.approx(float(ologs[k]), rel=2e-3), k
    assert float(res.loss) == pytest.approx(float(loss.detach()), rel=2e-3)
    _check_all_gradients(m, o, 4e-2, "cfg2_vits_b16_grad_report.json", {"gemm_calls": len(gemms), "gemm256_calls": len(big)})


@pytest.mark.parametrize("n_reg", [0, 4])
def test_cfg5_vit_large14_swiglu_518_step_matches_oracle(n_reg):
    """BASELINE configs[4] shapes, depth cut to 4: ViT-L/14 (D=1024, 16 heads), SwiGLU-fused FFN (hidden 2736: K = 2736 and N = 5472 --
    the partial-K-tile variant of the four-phase GEMM), 2 x 518^2 global crops (1370 tokens; 1374 with 4 register tokens and the
    antialiased, offset-free positional interpolation of the reg4 models) + 8 x 98^2 local crops, iBOT on, K = 65 536, batch 2
    => 5480 / 5496 global rows (>= 2048: the 256-row kernels).  Loss terms, gradient norm and EVERY parameter tensor's gradient
    against the fp32 oracle; the spies assert that the partial-K-tile kernel and the 1370-token attention calls happened."""
    import lightly_train_amd  # noqa: F401
    from lightly_train_amd import ops
    from lightly_train_amd.dinov2 import DINOv2, DINOv2Args, init_head_state
    from lightly_train_amd.vit import ViTConfig, init_vit_state
    from oracle import dinov2_oracle as O

    g = torch.Generator().manual_seed(61 + n_reg)
    extra = dict(num_register_tokens=4, interpolate_offset=0.0, interpolate_antialias=True) if n_reg else {}
    vc = ViTConfig(embed_dim=1024, depth=4, num_heads=16, mlp_ratio=4.0, patch_size=14, img_size=518, init_values=1e-5,
                   ffn_layer="swiglufused", **extra)
    bsd = init_vit_state(vc, g)
    assert bsd["blocks.0.mlp.w12.weight"].shape == (5472, 1024) and bsd["blocks.0.mlp.w3.weight"].shape == (1024, 2736)
    shs, ths = init_head_state(1024, 2048, 256, 65536, g), init_head_state(1024, 2048, 256, 65536, g)
    b = 2
    m = DINOv2(vc, DINOv2Args(koleo_loss_weight=0.0), global_batch_size=b, total_steps=100, device="cuda", backbone_state=bsd,
               student_head_state=shs, teacher_head_state=ths)
    ocfg = dict(patch_size=14, num_heads=16, depth=4)
    if n_reg:
        ocfg.update(interpolate_offset=0.0, interpolate_antialias=True)
    o = _oracle_to_gpu(O.OracleDINOv2(bsd, shs, ocfg, args=dict(koleo_loss_weight=0.0), global_batch_size=b, total_steps=100, teacher_head=ths))
    views = [torch.randn(b, 3, 518, 518, generator=g) for _ in range(2)] + [torch.randn(b, 3, 98, 98, generator=g) for _ in range(8)]
    random.seed(9)
    gemms, attn, undo = _install_spies()
    ovf = ops.reduce_overflows()
    try:
        res = m.training_step_impl({"views": views}, 0)
    finally:
        undo()
    torch.cuda.synchronize()
    assert ops.reduce_overflows() == ovf, "reduction ledger scratch exhausted at ViT-L widths (the step would fall back to atomics)"
    ntok = 1370 + n_reg
    assert ("fwd", ntok, 16) in attn and ("bwd", ntok, 16) in attn and ("fwd", 50 + n_reg, 16) in attn
    kt = [c for c in gemms if c[1] == "gemm256q_ktail"]
    # w3 forward (K = 2736) and the w12 dgrad (contraction over 5472) of the global crops, teacher and student
    assert {(c[0], c[4]) for c in kt} >= {("fwd", 2736), ("dgrad", 5472)}, sorted({(c[0], c[3], c[4]) for c in kt})
    tok = [c for c in gemms if c[0] in ("fwd", "dgrad") and c[2] == 2 * b * ntok]
    assert tok and all(c[1] in ("gemm256", "gemm256q_ktail") for c in tok), [c for c in tok if c[1] == "gemm128"][:3]

    masks = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in m._last_masks.items()}
    loss, ologs = o.forward_loss([v.cuda() for v in views], masks)
    loss.backward()
    logs = {k.split("/")[-1]: float(v) for k, v in res.log_dict.items()}
    for k in ("dino_global_loss", "dino_local_loss", "ibot_loss"):
        assert logs[k] == pytest.approx(float(ologs[k]), rel=2e-3), k
    assert float(res.loss) == pytest.approx(float(loss.detach()), rel=2e-3)
    _check_all_gradients(m, o, 4e-2, f"cfg5_vitl14_reg{n_reg}_grad_report.json",
                         {"ktail_calls": len(kt), "attn_calls": sorted(set(attn))})


def _well_conditioned_resnet50_state(cfg, g):
    """A resnet50 state at which fp32 and bf16 runs of the network are comparable tensor by tensor.  torchvision's default init is
    not such a state: with every BatchNorm at (1, 0) the 16 residual branches are as large as the identity paths, the activations
    decorrelate under any rounding within a few blocks (torch's own bf16 autocast is ~40 % off per gradient tensor there).
    Trained networks are not like that -- their branches are corrections to the identity path.  The state used here is torchvision's
    `zero_init_residual` idea stopped short of zero: the last BatchNorm of every bottleneck at gamma = 0.2 (+ jitter), every other
    BatchNorm affine jittered away from (1, 0) so that all terms of the backward are exercised."""
    from lightly_train_amd.resnet import init_resnet_state

    sd = init_resnet_state(cfg, g)
    for k in sd:
        if k.endswith(("running_mean", "running_var", "num_batches_tracked")) or not (".bn" in k or k.startswith("bn") or "downsample.1" in k):
            continue
        if k.endswith("bn3.weight"):
            sd[k] = 0.2 + 0.02 * torch.randn(sd[k].shape, generator=g)
        elif k.endswith("weight"):
            sd[k] = 1.0 + 0.1 * torch.randn(sd[k].shape, generator=g)
        else:
            sd[k] = 0.1 * torch.randn(sd[k].shape, generator=g)
    return sd


def test_cfg4_vitl16_teacher_to_resnet50_step_at_real_widths():
    """BASELINE configs[3] at its real widths: frozen DINOv3 ViT-L/16 teacher (D=1024, 16 heads, 4 storage tokens, RoPE; depth cut to
    2) -> the FULL torchvision resnet50 student (3-4-6-3 bottlenecks, 25.6 M parameters, 53 training-mode BatchNorms), one 224^2 view,
    batch 32 (BatchNorm statistics over 32 x 112^2 ... 32 x 7^2 elements), queue 8192 -- the kernels
    `bench.py --method distillationv3 --student resnet50` dispatches (7 x 7 x 2048 map -> 14 x 14 teacher grid resize, 196-token
    similarity GEMMs at D = 1024) -- against the restated torchvision module under autograd, run on the GPU twice: in fp32 (the oracle)
    and under `torch.autocast(bfloat16)` (= the reference's own `precision="bf16-mixed"` arithmetic), both from the state of
    `_well_conditioned_resnet50_state`.

    What holds tightly, and is asserted: both loss terms (1e-3; observed 4e-5), the projection heads' gradients (1e-2; observed
    3e-3), the gradients of the last BatchNorm (3e-2), i.e. everything up to the first BatchNorm *backward*.
    Where it breaks, and why the bound changes there: the first convolution weight behind that BatchNorm's backward is already 13 % off
    (layer4.2.conv3), layer1 43 % -- and torch's own bf16 autocast of the same module is off from its fp32 run by the same amount
    TENSOR BY TENSOR (0.1331 vs 0.1357, 0.2766 vs 0.2819, 0.4294 vs 0.4294 ...; the two bf16 runs differ from each other by
    sqrt(2) x that: independent noise of equal size, not a shared bias).  Activations stored in bf16 move ReLU / max-pool decisions
    for the elements whose pre-activation straddles zero, a flipped element's gradient is wrong by its full magnitude, and a flip
    rate of 1-2 % per layer is a 10-15 % Frobenius error per layer: a property of bf16 activation storage -- which the reference's
    `precision="bf16-mixed"` shares -- not of a kernel (those are checked per op against torch at <= 1e-2, tests/test_gpu_ops.py, and the
    engine's orchestration in exact arithmetic, tests/test_resnet_engine_cpu.py).  Hence the bound for the convolutional trunk is
    relative to that yardstick: every tensor's error against fp32 must be <= 1.25 x the autocast run's error + 2e-2, and the median
    over the trunk <= the autocast run's median x 1.1 (observed 0.3729 vs 0.3716 .. 0.3758; the loss terms of the HIP step are closer
    to fp32 than the autocast run's: 7.0011 / 7.0244 against 7.0014).
    The report (all 161 + 4 tensors, both columns, per-stage medians) goes to gpurun_out/cfg4_resnet50_grad_report.json."""
    import statistics
    import lightly_train_amd  # noqa: F401
    from lightly_train_amd.dinov3 import dinov3_vit_config, export_dinov3_state
    from lightly_train_amd.distillationv3 import DistillationV3, DistillationV3Args
    from lightly_train_amd.resnet import ResNetConfig, from_flat_layout
    from lightly_train_amd.vit import init_vit_state
    from oracle import distill_oracle as OD

    g = torch.Generator().manual_seed(71)
    scfg = ResNetConfig()
    tcfg = dinov3_vit_config(1024, 2, 16, patch_size=16, img_size=224, layerscale_init=0.5)
    sd = _well_conditioned_resnet50_state(scfg, g)
    tsd = init_vit_state(tcfg, g)
    tsd["pos_embed"] = torch.zeros_like(tsd["pos_embed"])
    tsd["cls_token"] = torch.randn(tsd["cls_token"].shape, generator=g) * 0.02
    tsd["register_tokens"] = torch.randn(tsd["register_tokens"].shape, generator=g) * 0.02
    for k in tsd:
        if k.endswith("attn.qkv.bias"):
            tsd[k] = 0.02 * torch.randn(tsd[k].shape, generator=g)
            tsd[k][1024:2048] = 0
    Dt, Ds = 1024, 2048
    pg = {"weight": torch.nn.init.trunc_normal_(torch.empty(Dt, Ds), std=0.02, generator=g), "bias": torch.empty(Dt).uniform_(-0.02, 0.02, generator=g)}
    pl = {"weight": torch.nn.init.trunc_normal_(torch.empty(Dt, Ds), std=0.02, generator=g), "bias": torch.empty(Dt).uniform_(-0.02, 0.02, generator=g)}
    b, Q = 32, 8192
    m = DistillationV3(scfg, tcfg, DistillationV3Args(queue_size=Q), global_batch_size=b, total_steps=100, max_epochs=1, device="cuda",
                       student_state=sd, teacher_state=tsd, proj_global_state=pg, proj_local_state=pl)
    ocfg_s = dict(kind="resnet", layers=list(scfg.layers), width=scfg.width)
    ocfg_t = dict(patch_size=16, num_heads=16, depth=2, rope_base=100.0, ln_eps=1e-5)

    def gpu_oracle():
        o = OD.OracleDistillationV3(sd, ocfg_s, export_dinov3_state(tsd, tcfg), ocfg_t, pg, pl, Q, b, 100, weight_decay=1e-6)
        o.resnet.cuda()
        o.sb = {n: p_ for n, p_ in o.resnet.named_parameters() if not n.startswith("fc.")}
        for d in (o.pg, o.pl):
            for k in list(d):
                d[k] = d[k].detach().cuda().requires_grad_(True)
        o.teacher = {k: v.cuda() for k, v in o.teacher.items()}
        o.queue = o.queue.cuda()
        return o

    x = torch.randn(b, 3, 224, 224, generator=g)
    torch.manual_seed(300)
    res = m.training_step_impl({"views": [x]}, 0)
    torch.cuda.synchronize()
    lam, index = m._last["lam"], m._last["index"]
    o = gpu_oracle()
    loss, ologs = o.forward_loss(x.cuda(), lam, index.cuda())
    loss.backward()
    oa = gpu_oracle()      # the same module under the reference's mixed-precision mode
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss_a, alogs = oa.forward_loss(x.cuda(), lam, index.cuda())
    loss_a.backward()
    logs = {k.split("/")[-1]: float(v) for k, v in res.log_dict.items()}

    def grads(oo, n):
        if n.startswith("backbone."):
            return oo.sb[n[9:]].grad
        return (oo.pg[n[12:]] if n.startswith("proj_global.") else oo.pl[n[11:]]).grad

    ours_err, auto_err, ours_vs_auto = {}, {}, {}
    for n in m.student.names:
        ref = grads(o, n)
        mine = from_flat_layout(n[9:], m.student.g[n]) if n.startswith("backbone.") else m.student.g[n]
        ours_err[n] = fro(mine, ref)
        auto_err[n] = fro(grads(oa, n), ref)
        ours_vs_auto[n] = fro(mine, grads(oa, n))
    trunk = [n for n in ours_err if n.startswith("backbone.")]
    stages = {}
    for st in ("layer4", "layer3", "layer2", "layer1"):
        ns = [n for n in trunk if n.startswith("backbone." + st)]
        stages[st] = (statistics.median(ours_err[n] for n in ns), statistics.median(auto_err[n] for n in ns))
    med_o, med_a = statistics.median(ours_err[n] for n in trunk), statistics.median(auto_err[n] for n in trunk)
    out_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "cfg4_resnet50_grad_report.json"), "w") as f:
            json.dump({"columns": "relative Frobenius error: [HIP step vs fp32 oracle, torch bf16 autocast of the oracle module vs fp32 oracle, HIP step vs that autocast run]",
                       "trunk_median": [med_o, med_a], "per_stage_median": stages, "logs": logs, "oracle_fp32": ologs, "oracle_autocast": alogs,
                       "trunk_median_hip_vs_autocast": statistics.median(ours_vs_auto[n] for n in trunk),
                       "per_tensor": {n: [ours_err[n], auto_err[n], ours_vs_auto[n]] for n in ours_err}}, f, indent=1)
    assert logs["global_loss"] == pytest.approx(ologs["global_loss"], rel=1e-3)
    assert logs["local_loss"] == pytest.approx(ologs["local_loss"], rel=1e-3)
    for n in ("proj_global.weight", "proj_global.bias", "proj_local.weight", "proj_local.bias"):
        assert ours_err[n] < 1e-2, (n, ours_err[n])
    for n in ("backbone.layer4.2.bn3.weight", "backbone.layer4.2.bn3.bias"):
        assert ours_err[n] < 3e-2, (n, ours_err[n])
    # Per tensor: within 1.25 x the reference module's OWN bf16-autocast error of that tensor -- or of its stage's median tensor, whichever is
    # larger.  In the early stages both runs sit 35-45 % from the fp32 gradients (the signal has passed 16 BatchNorm'd blocks in bf16) and the
    # autocast run is itself a draw (torch's convolution backward is not reproducible run to run): a bound made of one tensor's single draw
    # failed by 4e-4 once in five full-suite runs of round 6 (backbone.layer1.2.bn1.bias: ours 0.4274, autocast 0.3256, stage median 0.4254).
    def bound(n):
        st_ = n.split(".")[1]
        return 1.25 * max(auto_err[n], stages[st_][1] if st_ in stages else 0.0) + 2e-2
    bad = [(n, ours_err[n], auto_err[n]) for n in trunk if not ours_err[n] <= bound(n)]
    assert not bad, sorted(bad, key=lambda t: t[2] - t[1])[:8]
    assert med_o <= 1.1 * med_a, (med_o, med_a)
