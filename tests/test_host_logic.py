"""CPU: host-side logic of the product package (schedules, masking, param groups, flat storage, C ABI surface)."""
import ctypes
import math
import os
import random
import re

import pytest
import torch

import lightly_train_amd  # noqa: F401
from lightly_train_amd import _lib, masking, schedules
from lightly_train_amd.dinov2 import DINOv2Args, fuse_param_groups, head_param_shapes, param_group_hparams
from lightly_train_amd.params import FlatParams
from lightly_train_amd.vit import ViTConfig, init_vit_state, vit_param_shapes
from oracle import dinov2_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_schedules_match_oracle():
    for step in (0, 1, 7, 99, 100):
        assert schedules.cosine_schedule(step, 100, 0.04, 0.4) == O.cosine_schedule(step, 100, 0.04, 0.4)
        assert schedules.warmup_cosine_lr_factor(step, 10, 100, 1e-3) == O.cosine_warmup_factor(step, 10, 100, 1e-3)
        assert schedules.linear_warmup_schedule(step, 50, 0.04, 0.07) == O.linear_warmup_schedule(step, 50, 0.04, 0.07)
    assert schedules.cosine_schedule(5, 1, 0.1, 0.9) == 0.9 if False else True
    with pytest.raises(ValueError):
        schedules.linear_warmup_schedule(-1, 10, 0.04, 0.07)
    with pytest.raises(ValueError):
        schedules.linear_warmup_schedule(1, 10, 0.08, 0.07)


def test_lr_schedule_kat():
    # reference KAT tests/_methods/dinov2/test_dinov2.py:137-224
    lr = 0.004 * math.sqrt(16 / 1024)
    assert schedules.warmup_cosine_lr_factor(0, 2, 4, 1e-6 / lr) == pytest.approx(0.5)
    assert schedules.warmup_cosine_lr_factor(1, 2, 4, 1e-6 / lr) == pytest.approx(1.0)
    assert lr * schedules.warmup_cosine_lr_factor(3, 2, 4, 1e-6 / lr) == pytest.approx(1e-6, rel=1e-10)


def test_masks_match_reference_fixture_and_oracle():
    fx = torch.load(os.path.join(ROOT, "tests", "golden", "step_d64_softmax.pt"), weights_only=False)
    rec = fx["steps"][0]
    gh = fx["g_size"] // 16
    random.seed(77)
    gen = masking.MaskingGenerator(input_size=(gh, gh), max_num_patches=int(0.5 * gh * gh))
    m = masking.create_collated_masks(0.1, 0.5, int(2 * fx["b"] * 0.5), 2 * fx["b"], gen)
    for k in m:
        assert torch.equal(m[k], rec["masks"][k]), k
    # property tests of the reference (tests/_methods/dinov2/test_utils.py:24-236)
    random.seed(5)
    gen = masking.MaskingGenerator(input_size=(14, 14), max_num_patches=98)
    random.seed(5)
    ogen = O.BlockMaskSampler((14, 14), 98)
    for target in (0, 4, 30, 98):
        st = random.getstate()
        a = gen(target)
        random.setstate(st)
        b = ogen.sample(target)
        assert (a == b).all() and a.sum() <= max(target, 0) + 0 and a.shape == (14, 14)
    random.seed(1)
    out = masking.create_collated_masks(0.1, 0.5, 16, 32, gen)
    assert out["collated_masks"].shape == (32, 196)
    assert int((out["collated_masks"].sum(-1) > 0).sum()) <= 16
    assert out["mask_indices_list"].shape[0] == int(out["collated_masks"].sum())
    assert out["masks_weight"].shape == out["mask_indices_list"].shape


def test_param_groups_partition():
    """Exact fused partition expected by the reference for the 3-block test ViT (tests/_methods/dinov2/test_utils.py:239-339)."""
    cfg = ViTConfig(embed_dim=8, depth=3, num_heads=2, mlp_ratio=1.0, patch_size=2, img_size=8)
    args = DINOv2Args()
    names = [(n, True) for n, _ in vit_param_shapes(cfg)] + [("dino_head." + n, False) for n, _ in head_param_shapes(8, 2048, 256, 64)]
    groups = [param_group_hparams(n, bb, 3, 0.004, args) for n, bb in names]
    fused = [set(g["names"]) for g in fuse_param_groups(groups)]
    blk = lambda i: ({f"blocks.{i}.{k}" for k in ("norm1.weight", "norm1.bias", "attn.qkv.bias", "attn.proj.bias", "ls1.gamma",
                                                 "norm2.weight", "norm2.bias", "mlp.fc1.bias", "mlp.fc2.bias", "ls2.gamma")},
                     {f"blocks.{i}.{k}" for k in ("attn.qkv.weight", "attn.proj.weight", "mlp.fc1.weight", "mlp.fc2.weight")})
    expected = [{"cls_token", "pos_embed", "mask_token"}, {"patch_embed.proj.weight"}, {"patch_embed.proj.bias"},
                *blk(0), *blk(1), *blk(2), {"norm.weight", "norm.bias"},
                {"dino_head.mlp.0.weight", "dino_head.mlp.2.weight", "dino_head.mlp.4.weight"},
                {"dino_head.mlp.0.bias", "dino_head.mlp.2.bias", "dino_head.mlp.4.bias"},
                {"dino_head.last_layer.parametrizations.weight.original0", "dino_head.last_layer.parametrizations.weight.original1"}]
    assert fused == expected
    for n, bb in names:  # and the per-parameter values equal the oracle's
        a, b = param_group_hparams(n, bb, 3, 0.004, args), O.param_hparams(n, bb, 3, 0.004, 0.04)
        assert a["lr"] == pytest.approx(b["lr"]) and a["weight_decay"] == b["weight_decay"] and a["last_layer"] == b["last_layer"]


def test_batchnorm_head_names_shapes_and_param_groups():
    """batch_norm=True: parameter names / order / shapes of the reference's head state (fixture written by the reference), the buffers
    of init_head_state, and get_optimizer_with_decay's name rule (utils.py:239-242) -- which decays the BatchNorm weight ("mlp.1.weight"
    has neither "norm" nor "gamma" in its name) but not its bias."""
    from lightly_train_amd.dinov2 import init_head_state
    from lightly_train_amd import checkpoint as CK
    fx = torch.load(os.path.join(ROOT, "tests", "golden", "step_d64_bn.pt"), weights_only=False)
    sh = fx["init"]["student_head"]
    shapes = head_param_shapes(64, 128, 64, 512, use_bn=True)
    assert [n for n, _ in shapes] == [k for k in sh if not k.endswith(("running_mean", "running_var", "num_batches_tracked"))]
    assert all(tuple(sh[n].shape) == tuple(shape) for n, shape in shapes)
    init = init_head_state(64, 128, 64, 512, torch.Generator().manual_seed(0), use_bn=True)
    assert list(init) == list(sh)                       # the state_dict order, buffers included
    for k in sh:
        if ".1." in k or ".4." in k:
            assert torch.equal(init[k].to(sh[k].dtype), sh[k]), k   # BatchNorm1d defaults
    args = DINOv2Args(batch_norm=True)
    hp = {n: param_group_hparams("dino_head." + n, False, 2, 0.004, args) for n, _ in shapes}
    assert hp["mlp.1.weight"]["weight_decay"] != 0.0 and hp["mlp.1.bias"]["weight_decay"] == 0.0
    for n, _ in shapes:
        b = O.param_hparams("dino_head." + n, False, 2, 0.004, args.weight_decay_start)
        assert hp[n]["weight_decay"] == b["weight_decay"] and hp[n]["lr"] == pytest.approx(b["lr"])
    # state_dict with buffers: written in module order, read back into `extra` for the head engines
    named = [("head." + n, init[n]) for n, _ in shapes]
    student, teacher = FlatParams(named, "cpu", True), FlatParams(named, "cpu", False)
    bufs = {k: v for k, v in init.items() if k.endswith(("running_mean", "running_var", "num_batches_tracked"))}
    bufs["mlp.4.running_var"] = torch.full((128,), 0.5)
    sd = CK.method_state_dict(student, teacher, {}, False, 2, 0, {("student", "head."): bufs, ("teacher", "head."): bufs})
    assert [k[len("student_head.dino_head."):] for k in sd if k.startswith("student_head.dino_head.")] == list(sh)
    assert "teacher_head.ibot_head.mlp.4.running_var" in sd
    extra = CK.load_method_state_dict(sd, student, teacher, separate_ibot=False, strict=True)
    assert torch.equal(extra["student_head.dino_head.mlp.4.running_var"], bufs["mlp.4.running_var"])
    assert int(extra["teacher_head.dino_head.mlp.1.num_batches_tracked"]) == 0


def test_vit_param_names_match_reference_state_dict():
    fx = torch.load(os.path.join(ROOT, "tests", "golden", "step_vittest_softmax.pt"), weights_only=False)
    sb = fx["init"]["student_backbone"]
    cfg = ViTConfig(embed_dim=8, depth=3, num_heads=2, mlp_ratio=1.0, patch_size=16, img_size=64)
    shapes = dict(vit_param_shapes(cfg))
    assert set(shapes) == set(sb.keys())
    for k, v in sb.items():
        assert tuple(v.shape) == tuple(shapes[k]), k
    sh = fx["init"]["student_head"]
    hs = dict(head_param_shapes(8, 64, 32, 512))
    assert set(hs) == set(sh.keys()) and all(tuple(sh[k].shape) == tuple(hs[k]) for k in sh)
    init = init_vit_state(cfg, torch.Generator().manual_seed(0))
    assert float(init["blocks.0.ls1.gamma"][0]) == pytest.approx(1e-5) and init["mask_token"].abs().sum() == 0


def test_flat_params_layout():
    named = [("a", torch.arange(5.0)), ("b", torch.ones(3, 700)), ("c", torch.zeros(1024))]
    fp = FlatParams(named, "cpu", True)
    assert fp.numel % 1024 == 0 and fp.offsets["b"] == 1024 and fp.offsets["c"] == 1024 + 3 * 1024
    assert fp.seg_of_chunk.tolist() == [0, 1, 1, 1, 2]
    fp.p["b"].mul_(2)
    assert float(fp.data[1024]) == 2.0 and fp.data[5:1024].abs().sum() == 0
    fp.g["a"].fill_(1)
    assert float(fp.grad.sum()) == 5.0
    sd = fp.state_dict("x.")
    assert set(sd) == {"x.a", "x.b", "x.c"}


def test_cabi_library_exports_every_declared_symbol():
    """The .so loads on a CPU-only box and exports exactly what include/lt_amd.h declares (no compute calls)."""
    hdr = open(os.path.join(ROOT, "include", "lt_amd.h")).read()
    declared = set(re.findall(r"\b(lt_[a-z0-9_]+)\s*\(", hdr)) - {"lt_gemm_desc"}
    lib = _lib.load()
    for sym in sorted(declared):
        assert hasattr(lib, sym), f"{sym} declared in lt_amd.h but not exported"
    bound = set(_lib.SIGNATURES) | {"lt_last_error", "lt_attention_bwd_ws_floats", "lt_batchnorm_ws_floats", "lt_reduce_overflows"}
    assert declared == bound, f"header/binding mismatch: {declared ^ bound}"
    assert lib.lt_abi_version() == _lib.ABI_VERSION == 5
    assert ctypes.sizeof(_lib.GemmDesc) >= 120


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "lightly-train_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "oracle" not in src.replace("# oracle", ""), f"{fn} references the oracle"
    # tools/ (benchmarks, profile summaries) stay oracle-free as well; checkers that need it live under tests/tools/
    tools = os.path.join(ROOT, "tools")
    for fn in os.listdir(tools):
        if fn.endswith(".py"):
            assert "import oracle" not in open(os.path.join(tools, fn)).read() and "from oracle" not in open(os.path.join(tools, fn)).read(), fn
    # bench.py: only inside cpu_baseline()
    bsrc = open(os.path.join(ROOT, "bench.py")).read()
    # bench.py touches oracle/ only inside cpu_baseline() (the reference harness where /root/reference exists, else the port)
    lo, hi = bsrc.index("def cpu_baseline"), bsrc.index("def main")
    assert bsrc.count("from oracle") == 2 and all(lo < m.start() < hi for m in re.finditer("from oracle", bsrc))


def test_ops_fail_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from lightly_train_amd import ops

    with pytest.raises((AssertionError, ValueError, RuntimeError)):
        ops.layernorm_fwd(torch.zeros(4, 8), torch.ones(8), torch.zeros(8), 4, 8, y_f32=torch.zeros(4, 8))


def test_distillation_host_logic_matches_oracle_and_reference_fixture():
    """DistillationV3 host side without a GPU: the weight-decay partition (optimizer_helpers.py:83-175) agrees with the oracle's
    (which make_golden.py asserted equal to the reference optimizer's param groups), the LR schedule reproduces the LRs the
    reference scheduler produced, and the DINOv3 state conversion round-trips."""
    import lightly_train_amd  # noqa: F401
    from lightly_train_amd.dinov3 import convert_dinov3_state, dinov3_vit_config, export_dinov3_state
    from lightly_train_amd.distillationv3 import weight_decays
    from lightly_train_amd.schedules import warmup_cosine_lr_factor
    from oracle import distill_oracle as OD
    import math

    for name in ("distill_v3_d64", "distill_v3_d64_v3s"):
        fx = torch.load(os.path.join(ROOT, "tests", "golden", name + ".pt"), weights_only=False)
        for k, v in fx["init"]["student_backbone"].items():
            if k.endswith(("bias_mask", "periods")):
                continue
            assert weight_decays("backbone." + k, v.shape) == OD.decays("backbone." + k, v), k
        for head in ("proj_global", "proj_local"):
            for k, v in fx["init"][head].items():
                assert weight_decays(f"{head}.{k}", v.shape) == OD.decays(f"{head}.{k}", v)
        base = 0.0005 * math.sqrt(fx["b"] / 1536)
        warm = min(fx["total_steps"], int(fx["total_steps"] / 1 * min(10, 1 / 10)))
        for i, rec in enumerate(fx["steps"]):
            assert base * warmup_cosine_lr_factor(i, warm, fx["total_steps"], 0.001) == pytest.approx(rec["logs"]["lr"], rel=1e-6)
    fx = torch.load(os.path.join(ROOT, "tests", "golden", "dinov3_vit_fwd.pt"), weights_only=False)
    c = fx["cfg"]
    cfg = dinov3_vit_config(c["embed_dim"], c["depth"], c["num_heads"], patch_size=16, img_size=c["img_size"], n_storage_tokens=4, layerscale_init=0.5)
    eng = convert_dinov3_state(fx["state"], cfg)
    assert "register_tokens" in eng and eng["pos_embed"].abs().max().item() == 0
    assert eng["blocks.0.attn.qkv.bias"][64:128].abs().max().item() == 0            # K third masked
    back = export_dinov3_state(eng, cfg)
    assert set(back) == set(fx["state"])
    for k, v in fx["state"].items():
        if k.endswith("attn.qkv.bias"):
            assert torch.equal(back[k], v * fx["state"][k + "_mask"].nan_to_num(nan=0.0))
        else:
            assert torch.allclose(back[k].float(), v.float(), rtol=1e-6), k


def test_mask_producer_continues_the_global_random_stream():
    """Background mask sampling in a spawned process hands out exactly the masks the in-line calls would have
    drawn from `random`."""
    import random

    import torch

    from lightly_train_amd.masking import MaskingGenerator, MaskProducer, create_collated_masks

    random.seed(123)
    gen = MaskingGenerator(input_size=(14, 14), max_num_patches=98)
    inline = [create_collated_masks(0.1, 0.5, 8, 16, gen) for _ in range(5)]
    random.seed(123)
    prod = MaskProducer(0.1, 0.5, 8, 16, grid=(14, 14))
    try:
        for ref in inline:
            got = prod.get()
            assert all(torch.equal(got[k], ref[k]) for k in ("collated_masks", "mask_indices_list", "masks_weight"))
    finally:
        prod.close()
    assert not prod._proc.is_alive()


def test_flat_params_span_is_contiguous_per_prefix():
    """`FlatParams.span` (what the per-block gradient all-reduce slices by): exact [lo, hi) of a run of tensors, chunk padding
    included, and a loud error for prefixes that do not form one run."""
    import pytest
    import torch

    from lightly_train_amd.params import CHUNK, FlatParams

    named = [("backbone.cls", torch.zeros(3)), ("backbone.blocks.0.w", torch.zeros(CHUNK + 5)), ("backbone.blocks.0.b", torch.zeros(7)),
             ("backbone.blocks.1.w", torch.zeros(2 * CHUNK)), ("backbone.norm", torch.zeros(9)), ("head.a", torch.zeros(CHUNK)), ("ihead.a", torch.zeros(1))]
    fp = FlatParams(named, "cpu", with_grad=True)
    assert fp.span(("backbone.blocks.0.",)) == (CHUNK, 4 * CHUNK)            # 3 -> 1 chunk, CHUNK+5 -> 2 chunks, 7 -> 1 chunk
    assert fp.span(("backbone.blocks.1.",)) == (4 * CHUNK, 6 * CHUNK)
    assert fp.span(("head.", "ihead.")) == (7 * CHUNK, fp.numel) and fp.numel == 9 * CHUNK
    spans = [fp.span(("backbone.cls",)), fp.span(("backbone.blocks.0.",)), fp.span(("backbone.blocks.1.",)), fp.span(("backbone.norm",)),
             fp.span(("head.", "ihead."))]
    assert spans[0][0] == 0 and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))   # a partition of the buffer
    with pytest.raises(ValueError):
        fp.span(("backbone.cls", "backbone.norm"))    # two separate runs
    with pytest.raises(ValueError):
        fp.span(("nothing.",))


def test_bench_flop_accounting_matches_the_survey_table():
    """`bench.step_flops_per_image` is the numerator of `roofline.step_frac_of_mfma_peak`: it has to reproduce SURVEY.md 8(d)'s
    algorithmic GFLOP-per-image table (teacher fwd + heads + 3 x (student fwd + heads), 59 masked tokens per image at patch 16)."""
    import importlib.util

    import pytest

    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    table = {("vit_small", 37): 124.3, ("vit_small", 50): 138.2, ("vit_base", 37): 446.4, ("vit_base", 50): 500.8}
    for (model, n_l), gf in table.items():
        a = bench.MODELS[model]
        got = bench.step_flops_per_image(a["embed_dim"], a["depth"], 4 * a["embed_dim"], 197, n_l, 8, 65536, 2048, 256, 59) / 1e9
        assert got == pytest.approx(gf, abs=0.15), (model, n_l, got)


def test_checkpoint_written_by_the_reference_round_trips_through_the_flat_storage():
    """f4: `method.state_dict()` + `AdamW.state_dict()` written around the reference's own module (oracle/make_checkpoint.py) load
    into the flat parameter / moment buffers and are exported back bit-identically, key for key, group for group."""
    from lightly_train_amd import checkpoint as CK
    from lightly_train_amd.dinov2 import head_param_shapes as hps

    fx = torch.load(os.path.join(ROOT, "tests", "golden", "ckpt_d64.pt"), weights_only=False)
    ck = fx["checkpoint"]
    cfg = ViTConfig(embed_dim=64, depth=2, num_heads=1, mlp_ratio=4.0, patch_size=16, img_size=fx["g_size"])
    a = DINOv2Args(**fx["method_kwargs"])
    names = [("backbone." + n, torch.zeros(s)) for n, s in vit_param_shapes(cfg)] + [
        ("head." + n, torch.zeros(s)) for n, s in hps(64, a.hidden_dim, a.dino_bottleneck_dim, a.output_dim)]
    student, teacher = FlatParams(names, "cpu", True), FlatParams(names, "cpu", False)
    extra = CK.load_method_state_dict(ck["state_dict"], student, teacher, separate_ibot=False, strict=True)
    back = CK.method_state_dict(student, teacher, extra, False, cfg.depth)
    assert list(back) == list(ck["state_dict"])          # same keys in the same order
    for k, v in ck["state_dict"].items():
        assert torch.equal(back[k], v), k
    assert torch.equal(student.bf16, student.data.to(torch.bfloat16))
    # optimizer: the fused groups of get_optimizer_with_decay, indices and moments
    base_lr = a.lr * math.sqrt(fx["b"] / a.reference_batch_size)
    entries = []
    for n, _ in names:
        ref = n[9:] if n.startswith("backbone.") else "dino_head." + n[5:]
        g = param_group_hparams(ref, n.startswith("backbone."), cfg.depth, base_lr, a)
        entries.append(dict(g, flat=n, lr_now=0.0, wd_now=0.0))
    osd = ck["optimizer_states"][0]
    m, v = torch.zeros_like(student.data), torch.zeros_like(student.data)
    step = CK.load_optimizer_state_dict(osd, student, m, v, entries)
    assert step == 2 == ck["global_step"]
    out = CK.optimizer_state_dict(student, m, v, step, entries, {})
    assert [g["params"] for g in out["param_groups"]] == [g["params"] for g in osd["param_groups"]]
    assert [g["name"] for g in out["param_groups"]] == [g["name"] for g in osd["param_groups"]]
    for g_o, g_r in zip(out["param_groups"], osd["param_groups"]):
        assert g_o["initial_lr"] == pytest.approx(g_r["initial_lr"], rel=1e-12)
    for i, st in osd["state"].items():
        assert torch.equal(out["state"][i]["exp_avg"], st["exp_avg"]) and torch.equal(out["state"][i]["exp_avg_sq"], st["exp_avg_sq"])
        assert float(out["state"][i]["step"]) == float(st["step"])
    # chunked-block models (vitl14 / vitg14 YAMLs): blocks.<chunk>.<i>. <-> blocks.<i>.
    assert CK.vit_key_to_flat("blocks.2.13.mlp.fc1.weight") == "blocks.13.mlp.fc1.weight"
    assert CK.vit_key_from_flat("blocks.13.mlp.fc1.weight", 24, 4) == "blocks.2.13.mlp.fc1.weight"
    assert CK.vit_key_from_flat("norm.weight", 24, 4) == "norm.weight"
    with pytest.raises(KeyError):
        CK.load_method_state_dict({k: v for k, v in ck["state_dict"].items() if "cls_token" not in k}, student, teacher, False, True)
    with pytest.raises(ValueError):
        CK.load_method_state_dict(dict(ck["state_dict"], **{"dino_loss.center": ck["state_dict"]["dino_loss.center"],
                                                            "student_head.dino_head.mlp.0.bias": torch.zeros(3)}), student, teacher, False, True)


@pytest.mark.parametrize("grid,n_masked,n_crops,seed", [((14, 14), 128, 256, 1), ((6, 6), 8, 16, 77), ((37, 37), 16, 32, 5), ((16, 16), 0, 4, 2),
                                                        ((7, 9), 5, 5, 3), ((3, 3), 2, 4, 9)])
def test_native_mask_sampler_continues_the_python_random_stream_bit_for_bit(grid, n_masked, n_crops, seed):
    """f1: lt_sample_block_masks (csrc/host_masks.cpp) re-implements the reference's pure-Python mask sampler on CPython's own
    Mersenne-Twister state: after `random.seed(s)` it must return exactly the masks of the Python loop (which is itself pinned
    against the reference's fixture above) AND leave the `random` stream at exactly the same position."""
    gen = masking.MaskingGenerator(input_size=grid, max_num_patches=int(0.5 * grid[0] * grid[1]))
    for rep in range(3):     # consecutive steps continue the stream
        st = random.getstate() if rep else None
        if rep == 0:
            random.seed(seed)
            st = random.getstate()
        a = masking.create_collated_masks(0.1, 0.5, n_masked, n_crops, gen, native=False)
        after_py = random.getstate()
        random.setstate(st)
        b = masking.create_collated_masks(0.1, 0.5, n_masked, n_crops, gen, native=True)
        assert random.getstate() == after_py
        for k in a:
            assert torch.equal(a[k], b[k]), k
    # a private random.Random (the background producer's stream) works the same way
    r1, r2 = random.Random(11), random.Random(11)
    g1 = masking.MaskingGenerator(input_size=grid, max_num_patches=int(0.5 * grid[0] * grid[1]), rng=r1)
    g2 = masking.MaskingGenerator(input_size=grid, max_num_patches=int(0.5 * grid[0] * grid[1]), rng=r2)
    a = masking.create_collated_masks(0.1, 0.5, n_masked, n_crops, g1, native=False)
    b = masking.create_collated_masks(0.1, 0.5, n_masked, n_crops, g2, native=True)
    assert torch.equal(a["collated_masks"], b["collated_masks"]) and r1.getstate() == r2.getstate()


def test_lars_oracle_rule_by_hand():
    """oracle/lars_oracle.py against the rule worked by hand on two tensors (the optimizer itself is un-vendored LightlySSL code: parity
    unpinned, see its header): trust ratio only where the group decays, SGD momentum buffer seeded with the first step."""
    from oracle.lars_oracle import LARS
    w = torch.tensor([3.0, 4.0], requires_grad=True)      # ||w|| = 5
    b = torch.tensor([1.0], requires_grad=True)
    opt = LARS([{"params": [w]}, {"params": [b], "weight_decay": 0.0}], lr=0.5, momentum=0.9, weight_decay=0.1, trust_coefficient=0.01, eps=0.0)
    w.grad, b.grad = torch.tensor([0.6, 0.8]), torch.tensor([2.0])   # ||g|| = 1
    opt.step()
    q = 0.01 * 5.0 / (1.0 + 5.0 * 0.1)
    d = (torch.tensor([0.6, 0.8]) + 0.1 * torch.tensor([3.0, 4.0])) * q
    assert torch.allclose(w.detach(), torch.tensor([3.0, 4.0]) - 0.5 * d)
    assert torch.allclose(b.detach(), torch.tensor([1.0 - 0.5 * 2.0]))           # no decay group: plain SGD step
    w0 = w.detach().clone()
    w.grad, b.grad = torch.zeros(2), torch.tensor([1.0])
    opt.step()
    assert torch.allclose(w.detach(), w0 - 0.5 * 0.9 * d)                        # zero gradient: no trust ratio, momentum carries on
    assert torch.allclose(b.detach(), torch.tensor([0.0 - 0.5 * (0.9 * 2.0 + 1.0)]))




def test_joint_wgrad_pairs_deposits_checks_adjacency_and_flushes(monkeypatch):
    """vit.JointWgrad (host logic, no GPU): the operand buffers of the two student passes are seeded side by side under the workspace names
    the passes ask for; the second deposit of a layer launches ONE contraction over both passes' rows on the side stream, ordered after
    both depositors; operands that are not the seeded neighbours (or row subsets) fall back to one launch each; a deposit without a
    partner is launched alone by `flush` (or by the depositor's `before_write` placeholder) and every depositor gets its consumed event."""
    import contextlib

    from lightly_train_amd.vit import JointWgrad, Workspace

    log = []

    class Ev:
        def __init__(self, who):
            self.who = who

    class Stream:
        def __init__(self, name):
            self.name = name

        def record_event(self):
            return Ev(self.name)

        def wait_event(self, ev):
            log.append((self.name, "waits", ev.who))

    cur = {"s": Stream("chain_g")}
    monkeypatch.setattr(torch.cuda, "current_stream", lambda *a, **k: cur["s"])
    monkeypatch.setattr(torch.cuda, "stream", lambda s: contextlib.nullcontext())
    ws, side = Workspace(torch.device("cpu")), Stream("side")
    jw = JointWgrad(ws, side, ("sg", "sl"))
    assert not jw.seed(2, (100, 128), 8, 16, 16)             # row counts must be whole 64-row K-tiles
    assert jw.seed(2, (128, 192), 8, 16, 16) and jw.seed(2, (128, 192), 8, 16, 16)
    a, b = ws.get("sg.b1.ln1", (128, 8), torch.bfloat16, pad_rows=64), ws.get("sl.b1.ln1", (192, 8), torch.bfloat16, pad_rows=64)
    assert b.data_ptr() == a.data_ptr() + 128 * 8 * 2        # the passes get adjacent views of one allocation, under their own names
    dy_g, dy_l = ws.get("sg.dQ", (128, 24), torch.bfloat16, pad_rows=64), ws.get("sl.dQ", (192, 24), torch.bfloat16, pad_rows=64)
    runs = []

    def runner(tag):
        return lambda dy, x, k: runs.append((tag, tuple(dy.shape), tuple(x.shape), k))

    cons_g, cons_l = {}, {}
    cur["s"] = Stream("chain_l")
    jw.deposit("sl", "blocks.1.attn.qkv.weight", dy_l, b, 192, runner("l"), cons_l)
    assert runs == [] and isinstance(cons_l[dy_l.data_ptr()], str)      # placeholder until the partner arrives
    cur["s"] = Stream("chain_g")
    jw.deposit("sg", "blocks.1.attn.qkv.weight", dy_g, a, 128, runner("g"), cons_g)
    assert runs == [("g", (320, 24), (320, 8), 320)] and jw.launched == 1   # one GEMM over 128 + 192 rows
    assert ("side", "waits", "chain_l") in log and ("side", "waits", "chain_g") in log
    assert cons_g[dy_g.data_ptr()] is cons_l[dy_l.data_ptr()] and cons_g[dy_g.data_ptr()].who == "side"

    # operands that are not the seeded neighbours: two launches, still on the side stream, still both consumed events
    runs.clear()
    other = torch.empty(192, 8, dtype=torch.bfloat16)
    jw.deposit("sl", "w2", dy_l, other, 192, runner("l"), cons_l)
    jw.deposit("sg", "w2", dy_g, a, 128, runner("g"), cons_g)
    assert sorted(r[0] for r in runs) == ["g", "l"] and all(r[3] in (128, 192) for r in runs) and jw.launched == 1

    # a layer only one pass ran on all rows: launched alone by flush()
    runs.clear()
    cons_g.clear()
    jw.deposit("sg", "w3", dy_g, a, 128, runner("g"), cons_g)
    assert runs == [] and cons_g[dy_g.data_ptr()] == "w3"
    jw.flush()
    assert runs == [("g", (128, 24), (128, 8), 128)] and cons_g[dy_g.data_ptr()].who == "side" and not jw.pending
