"""CPU, build container only (needs /root/reference): the executable drop-in `DINOv2AMD(Method)` of lightly_train_amd/integration.py at the
reference's own plugin point -- BASELINE configs[0] (plumbing: `_vittest14`, batch 16, 2 steps, accelerator = cpu).

The reference's `Method` base class, `method_helpers.get_method_cls`, its module containers, `Checkpoint` envelope and export path are the
REAL reference code, imported from /root/reference through oracle/ref_harness.py's stub Lightning (pytorch_lightning / lightly / cv2 ... are
not installable here); the HIP kernels are replaced by the plain-torch statements of their contracts (tests/tools/ops_emu.py, fp32), as in
tests/test_dinov2_method_cpu.py.  Checked:
  * the class is a `Method`, `get_method_cls(instance)` returns it, `install_as("dinov2")` maps the name to it;
  * two training steps driven in Lightning's hook order equal the reference's own `DINOv2` class on identical weights, views and mask
    draws: loss terms (3e-5; KoLeo, a nearest-neighbour distance of cls tokens that agree to 1e-6 at LayerScale 1e-5, at 2e-2),
    and with the KoLeo weight at 0 every student / EMA-teacher tensor after the two steps (3e-5 absolute: 6 % of one AdamW step of
    lr 5e-4 -- entries whose gradient is at round-off level take +-lr steps whose sign no two fp32 implementations share);
  * a checkpoint assembled the way Lightning + the reference's ModelCheckpoint callback assemble it (dump -> `on_save_checkpoint` -> the
    `lightly_train` envelope with the pickled containers) is read back by the reference's `Checkpoint.from_dict`, and
    `_commands/export.py::_get_model` returns modules whose weights are the trained EMA teacher -- bit-identical to the flat storage;
  * `on_load_checkpoint` resumes a fresh object from that file: its third step equals the original's third step.
"""
import os
import random
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))

from oracle import ref_harness as H  # noqa: E402

pytestmark = pytest.mark.skipif(not H.reference_available(), reason="needs the reference tree at /root/reference (build container only)")

import lightly_train_amd  # noqa: E402,F401
import ops_emu  # noqa: E402
from lightly_train_amd import ops  # noqa: E402
from test_dinov2_method_cpu import F32Workspace  # noqa: E402
from test_distillation_methods_cpu import _NoStream  # noqa: E402  (accepts the device= argument of torch.cuda.Stream)


@pytest.fixture(autouse=True)
def _no_cuda_streams(monkeypatch):
    import contextlib

    monkeypatch.setattr(torch.cuda, "current_stream", lambda *a, **k: _NoStream())
    monkeypatch.setattr(torch.cuda, "set_stream", lambda s: None)
    monkeypatch.setattr(torch.cuda, "Stream", _NoStream)
    monkeypatch.setattr(torch.cuda, "stream", lambda s: contextlib.nullcontext())


def views_for(step, b=16):
    g = torch.Generator().manual_seed(500 + step)
    return [torch.randn(b, 3, 224, 224, generator=g) for _ in range(2)] + [torch.randn(b, 3, 98, 98, generator=g) for _ in range(8)]


def exactify(m):
    """fp32 everywhere (see tests/test_dinov2_method_cpu.py::build_exact)."""
    m.ws = F32Workspace(torch.device("cpu"))
    for fp in (m.student, m.teacher):
        fp.bf16 = fp.data.clone()
        fp.b = {n: fp.bf16[fp.offsets[n]:fp.offsets[n] + fp.p[n].numel()].view(fp.shapes[n]) for n in fp.names}
    for h in {id(x): x for x in (m.s_head, m.t_head, m.s_ihead, m.t_ihead)}.values():
        h.wn = h.wn.float()
    for v in (m.s_vit, m.t_vit):
        if v.wpe_pad is not None:
            v.wpe_pad = v.wpe_pad.float()
    m._refresh_derived()


def build_pair(koleo, total_steps=2):
    from lightly_train_amd import integration

    H.install()
    kw = dict(arch="_vit_test", patch_size=14, img_size=224, global_batch_size=16, total_steps=total_steps, seed=0,
              method_kwargs=dict(koleo_loss_weight=koleo))
    ref = H.build_reference_method(**kw)
    cls = integration.dinov2_amd_method_cls()
    amd = H.build_reference_method(method_cls=cls, method_cls_kwargs=dict(device=torch.device("cpu")), **kw)
    return ref, amd, cls


def drive(amd, views, step):
    """What Lightning's loop does around a manual-optimization module for one batch."""
    batch = {"views": views, "filename": []}
    amd.trainer.global_step = step
    res = amd.training_step_impl(batch, step)
    amd.trainer.global_step = step + 1
    try:
        amd.on_train_batch_end(None, batch, step)
    except Exception:      # batch-timing hooks of the base class need a real Trainer
        pass
    out = {k.split("/")[-1]: float(v) for k, v in res.log_dict.items()}
    out["loss"] = float(res.loss)
    return out


def test_dinov2_amd_is_a_registered_reference_method():
    from lightly_train_amd import integration

    H.install()
    from lightly_train._methods import method_helpers
    from lightly_train._methods.dinov2.dinov2 import DINOv2Args
    from lightly_train._methods.method import Method

    _, amd, cls = build_pair(0.1)
    assert issubclass(cls, Method) and isinstance(amd, Method)
    assert method_helpers.get_method_cls(amd) is cls
    assert cls.method_args_cls() is DINOv2Args and cls.transform_cls().__name__ == "DINOv2ViTTransform"
    assert amd.automatic_optimization is False      # manual optimization: Lightning's DDP strategy leaves the wrapper's reducer off
    orig = method_helpers._method_name_to_cls
    try:
        integration.install_as("dinov2")
        assert method_helpers.get_method_cls("dinov2") is cls
        assert method_helpers.get_method_cls("dino").__name__ == "DINO"
        # gradient accumulation: the Trainer is built with 1 and carries k for OUR classes only; when the method that is resolved
        # afterwards (LT/_commands/train.py:433 then :476) is not one of them, Lightning's own window is restored
        from types import SimpleNamespace

        from lightly_train._commands import train_helpers

        wrapped = train_helpers.get_trainer
        inner = wrapped.__closure__[[c for c in wrapped.__code__.co_freevars].index("orig")]
        real = inner.cell_contents

        def fake_get_trainer(out, epochs, gradient_accumulation_steps, accelerator, strategy, devices, num_nodes, log_every_n_steps, precision, loggers,
                             callbacks, trainer_args):
            return SimpleNamespace(accumulate_grad_batches=gradient_accumulation_steps)

        inner.cell_contents = fake_get_trainer
        try:
            kw = dict(out=None, epochs=1, accelerator="cpu", strategy="auto", devices=1, num_nodes=1, log_every_n_steps=1, precision=None, loggers=[],
                      callbacks=[], trainer_args=None)
            tr = train_helpers.get_trainer(gradient_accumulation_steps=4, **kw)
            assert tr.accumulate_grad_batches == 1 and tr.lt_amd_accumulate_grad_batches == 4
            assert method_helpers.get_method_cls("dinov2") is cls
            assert tr.accumulate_grad_batches == 1 and tr.lt_amd_accumulate_grad_batches == 4       # ours: the binding accumulates
            tr = train_helpers.get_trainer(gradient_accumulation_steps=4, **kw)
            assert method_helpers.get_method_cls("dino").__name__ == "DINO"
            assert tr.accumulate_grad_batches == 4 and not hasattr(tr, "lt_amd_accumulate_grad_batches")   # not ours: Lightning accumulates
        finally:
            inner.cell_contents = real
    finally:
        method_helpers._method_name_to_cls = orig
    # identical containers: the state_dict keys of the drop-in are the reference's, in order
    ref, _, _ = build_pair(0.1)
    assert list(amd.state_dict()) == list(ref.state_dict())


@pytest.mark.parametrize("koleo", [0.1, 0.0])
def test_two_plumbing_steps_equal_the_reference_class(koleo):
    ref, amd, _ = build_pair(koleo)
    runner = H.ReferenceRunner(ref)
    with ops_emu.emulate(ops):
        exactify(amd.impl())
        for step in range(2):
            v = views_for(step)
            random.seed(70 + step); torch.manual_seed(70 + step)
            want = runner.train_step([x.clone() for x in v])
            random.seed(70 + step); torch.manual_seed(70 + step)
            got = drive(amd, v, step)
            for k in ("dino_global_loss", "dino_local_loss", "ibot_loss"):
                assert got[k] == pytest.approx(want[k], rel=3e-5, abs=3e-5), (step, k)
            assert got["koleo_loss"] == pytest.approx(want["koleo_loss"], rel=2e-2), step
            if koleo == 0.0:
                assert got["loss"] == pytest.approx(want["loss"], rel=3e-5)
        assert amd.impl().trainer.global_step == 2 and amd.impl().opt_step == 2
        # the scheduler entry of a checkpoint: the reference scheduler's own state_dict() after the same two steps, field by field
        want_s, got_s = runner.sched.state_dict(), amd.impl().lr_scheduler_state()
        assert set(got_s) >= set(want_s) - {"_is_initial"}, set(want_s) - set(got_s)
        for k in ("last_epoch", "_step_count", "warmup_epochs", "max_epochs"):
            assert got_s[k] == want_s[k], k
        assert got_s["end_value"] == pytest.approx(want_s["end_value"], rel=1e-12)
        assert got_s["base_lrs"] == pytest.approx(want_s["base_lrs"], rel=1e-9) and got_s["_last_lr"] == pytest.approx(want_s["_last_lr"], rel=1e-9)
        if koleo == 0.0:
            sd, rsd = amd.state_dict(), ref.state_dict()
            assert list(sd) == list(rsd)
            for k in rsd:
                assert torch.allclose(sd[k].float(), rsd[k].float(), atol=3e-5), (k, (sd[k].float() - rsd[k].float()).abs().max().item())


def test_checkpoint_envelope_is_read_and_exported_by_the_reference(tmp_path):
    H.install()
    import lightly_train
    if not hasattr(lightly_train, "__version__"):
        lightly_train.__version__ = "0.17.0"
    from lightly_train._checkpoint import CHECKPOINT_LIGHTLY_TRAIN_KEY, Checkpoint, CheckpointLightlyTrain, CheckpointLightlyTrainModels
    from lightly_train._commands import export as E
    from lightly_train._transforms.transform import NormalizeArgs

    ref, amd, _ = build_pair(0.0, total_steps=4)
    with ops_emu.emulate(ops):
        exactify(amd.impl())
        init_teacher = {k: v.clone() for k, v in amd.teacher_embedding_model.wrapped_model.get_model().state_dict().items()}
        for step in range(2):
            random.seed(80 + step); torch.manual_seed(80 + step)
            drive(amd, views_for(step), step)
        # --- what Trainer.save_checkpoint does: dump (stale module state), module hook, then the reference's ModelCheckpoint callback adds
        # the envelope around the containers it was given at set-up (the embedding model passed to the method = the EMA teacher)
        from lightly_train._methods.method import Method
        ckpt = {"epoch": 0, "global_step": 2, "state_dict": Method.state_dict(amd), "optimizer_states": [], "lr_schedulers": []}
        amd.on_save_checkpoint(ckpt)
        emb = amd.teacher_embedding_model
        env = CheckpointLightlyTrain.from_now(models=CheckpointLightlyTrainModels(model=emb.wrapped_model.get_model(), wrapped_model=emb.wrapped_model,
                                                                                   embedding_model=emb), normalize_args=NormalizeArgs())
        ckpt[CHECKPOINT_LIGHTLY_TRAIN_KEY] = env.to_dict()
        path = tmp_path / "last.ckpt"
        torch.save(ckpt, path)

        loaded = torch.load(path, weights_only=False)
        cp = Checkpoint.from_dict(loaded)                                   # the reference's reader
        model = E._get_model(checkpoint=cp, part=E.ModelPart.MODEL)         # what `lightly_train.export(part="model")` exports
        wrapped = E._get_model(checkpoint=cp, part=E.ModelPart.WRAPPED_MODEL)
        embm = E._get_model(checkpoint=cp, part=E.ModelPart.EMBEDDING_MODEL)
        want = amd.impl().export_backbone_state_dict()
        got = model.state_dict()
        assert list(got) == list(want)
        moved = 0
        for k in want:
            assert torch.equal(got[k], want[k].cpu()), k                     # bit-identical to the flat EMA-teacher storage
            moved += int(not torch.equal(got[k], init_teacher[k]))
        assert moved > 10                                                    # ... and trained: not the initial weights
        assert torch.equal(wrapped.get_model().state_dict()["cls_token"], want["cls_token"].cpu())
        assert torch.equal(embm.wrapped_model.get_model().state_dict()["norm.weight"], want["norm.weight"].cpu())
        # --- downstream: the file `lightly_train.export(format="package_default")` writes (DINOv2ViTPackage.export_model, dinov2_vit_package.py:
        # 146-162) loads strictly into a freshly built reference ViT of the same architecture -- what fine-tuning / inference start from -- and
        # that model's features equal the HIP engine's on the trained EMA-teacher weights
        from lightly_train._models import package_helpers
        from lightly_train._models.dinov2_vit.dinov2_vit_src.models import vision_transformer as vits

        package = package_helpers.get_package_from_model(model=model, include_custom=True, fallback_custom=True)
        out = tmp_path / "exported_model.pt"
        package.export_model(model=wrapped, out=out, log_example=False)
        fresh = getattr(vits, "_vit_test")(img_size=224, patch_size=14, init_values=1e-5, drop_path_rate=0.0, ffn_layer="mlp", block_chunks=0,
                                            interpolate_offset=0.1)
        fresh.load_state_dict(torch.load(out, weights_only=True), strict=True)
        fresh.eval()
        x = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(4))
        with torch.no_grad():
            ref_out = fresh(x, None, is_training=True)
        tctx = amd.impl().t_vit.forward(amd.impl().ws, "export_check", x, None, save=False)
        xn = tctx["xn"].view(2, -1, fresh.embed_dim)
        assert torch.allclose(xn[:, 0], ref_out["x_norm_clstoken"], atol=2e-5) and torch.allclose(xn[:, 1:], ref_out["x_norm_patchtokens"], atol=5e-5)
        # state_dict of the checkpoint = the module's keys with current values, optimizer state in torch.optim.AdamW's format
        assert list(cp.state_dict) == list(ref.state_dict())
        osd = loaded["optimizer_states"][0]
        assert set(osd) == {"state", "param_groups"} and all(int(s["step"]) == 2 for s in osd["state"].values())

        # --- resume: a fresh object loads the file, its third step equals the original's third step
        _, amd2, _ = build_pair(0.0, total_steps=4)
        amd2.on_load_checkpoint(loaded)
        # what Lightning does next with the SAME dict (checkpoint_connector.restore_optimizers -> strategy.load_optimizer_state_dict):
        # the optimizer `configure_optimizers` returned takes checkpoint["optimizer_states"][0] -- the hook above has swapped the
        # reference-format AdamW state (many parameter groups, consumed by the engine) for that optimizer's own
        amd2.configure_optimizers().load_state_dict(loaded["optimizer_states"][0])
        assert loaded["lr_schedulers"] == []
        assert amd2._pending_resume is not None and len(amd2._pending_resume["optimizer_states"][0]["param_groups"]) > 1
        exactify(amd2.impl())
        amd2.impl()._refresh_derived()
        v = views_for(2)
        # (the batch-center sums of the last step before the save are not part of a checkpoint -- `DINOLoss.async_batch_center` is a plain
        # attribute in the reference too, dinov2_loss.py:139-160 -- so a resumed run applies no center update at its first step; the
        # uninterrupted object is put into the same state for the comparison)
        amd.impl()._pending.clear()
        random.seed(90); torch.manual_seed(90)
        a = drive(amd, v, 2)
        random.seed(90); torch.manual_seed(90)
        b = drive(amd2, v, 2)
        assert b["loss"] == pytest.approx(a["loss"], rel=1e-6)
        s1, s2 = amd.state_dict(), amd2.state_dict()
        for k in s1:
            assert torch.allclose(s1[k].float(), s2[k].float(), atol=1e-7), k


def test_gradient_accumulation_equals_the_reference_under_accumulate_grad_batches():
    """`gradient_accumulation_steps = 2` (LT/_commands/train_helpers.py:224-236 -> Trainer(accumulate_grad_batches=2)): the reference class
    driven the way Lightning's automatic optimization drives it -- per micro-batch training_step, (loss / 2).backward(), the optimizer hooks
    and step on every second batch, `on_train_batch_end` (the EMA) after EVERY micro-batch -- against the binding accumulating two
    micro-batches of 8 in its flat gradient buffer: loss terms of all four micro-batches, and every student / EMA-teacher tensor after
    the two optimizer steps."""
    from lightly_train_amd import integration

    H.install()
    k, opt_steps = 2, 2
    kw = dict(arch="_vit_test", patch_size=14, img_size=224, global_batch_size=16, seed=0, method_kwargs=dict(koleo_loss_weight=0.0))
    ref = H.build_reference_method(total_steps=opt_steps, **kw)
    cls = integration.dinov2_amd_method_cls()
    amd = H.build_reference_method(method_cls=cls, method_cls_kwargs=dict(device=torch.device("cpu"), gradient_accumulation_steps=k),
                                   total_steps=opt_steps * k, **kw)   # a Trainer built with accumulate_grad_batches = 1 counts batches
    runner = H.ReferenceRunner(ref)
    with ops_emu.emulate(ops):
        assert integration.total_optimizer_steps(amd) == opt_steps
        exactify(amd.impl())
        assert amd.impl().trainer.estimated_stepping_batches == opt_steps
        for mb in range(opt_steps * k):
            v = views_for(mb, b=8)
            batch = {"views": [x.clone() for x in v], "filename": []}
            random.seed(170 + mb); torch.manual_seed(170 + mb)
            res = ref.training_step_impl(batch, mb)
            (res.loss / k).backward()
            if (mb + 1) % k == 0:
                ref.on_before_optimizer_step(runner.optim)
                torch.nn.utils.clip_grad_norm_([p for g in runner.optim.param_groups for p in g["params"]], ref.method_args.gradient_clip_val)
                runner.optim.step()
                runner.optim.zero_grad(set_to_none=True)
                runner.sched.step()
                ref.trainer.global_step += 1
            try:
                ref.on_train_batch_end(None, batch, mb)
            except Exception:
                pass
            want = {kk.split("/")[-1]: float(vv) for kk, vv in res.log_dict.items()}
            random.seed(170 + mb); torch.manual_seed(170 + mb)
            got_res = amd.training_step_impl({"views": v, "filename": []}, mb)
            if (mb + 1) % k == 0:
                amd.trainer.global_step += 1          # (Lightning: the progress-counter optimizer stepped)
            got = {kk.split("/")[-1]: float(vv) for kk, vv in got_res.log_dict.items()}
            for name in ("dino_global_loss", "dino_local_loss", "ibot_loss"):
                assert got[name] == pytest.approx(want[name], rel=3e-5, abs=3e-5), (mb, name)
            assert float(got_res.loss) == pytest.approx(float(res.loss), rel=3e-5), mb
        assert amd.impl().opt_step == opt_steps and amd.impl().trainer.global_step == opt_steps
        sd, rsd = amd.state_dict(), ref.state_dict()
        for name in rsd:
            assert torch.allclose(sd[name].float(), rsd[name].float(), atol=3e-5), (name, (sd[name].float() - rsd[name].float()).abs().max().item())


def test_optimizer_step_total_counts_the_window_that_closes_with_every_epoch():
    """A window also closes with the epoch's last batch (`begin_micro_batch`, like Lightning): 10 batches per epoch at k = 4 are 3 optimizer
    steps per epoch -- 300 over 100 epochs, not ceil(1000 / 4) = 250 (the schedules raise past their total).  `max_steps` caps it; without a
    per-epoch batch count the Trainer's estimate is used."""
    from types import SimpleNamespace

    from lightly_train_amd import integration

    def host(**tr):
        return SimpleNamespace(trainer=SimpleNamespace(**tr), gradient_accumulation_steps=1)

    h = host(estimated_stepping_batches=1000, num_training_batches=10, max_epochs=100, max_steps=-1, lt_amd_accumulate_grad_batches=4)
    assert integration.total_optimizer_steps(h) == 300
    # drive the window logic over one run and count the boundaries it reports
    class Eng:
        supports_accumulation = True
        trainer = SimpleNamespace(global_step=0)
    eng = Eng()
    h.impl, h._micro, h.trainer.global_step = (lambda: eng), 0, 0
    steps = 0
    for _ in range(100):
        for bi in range(10):
            h.trainer.is_last_batch = bi == 9
            _, boundary = integration.begin_micro_batch(h, bi)
            steps += int(boundary)
    assert steps == 300
    h.trainer.max_steps = 120
    assert integration.total_optimizer_steps(h) == 120
    assert integration.total_optimizer_steps(host(estimated_stepping_batches=1000, num_training_batches=12, max_epochs=100, max_steps=-1,
                                                  lt_amd_accumulate_grad_batches=4)) == 300
    assert integration.total_optimizer_steps(host(estimated_stepping_batches=1000, num_training_batches=float("inf"), max_epochs=-1, max_steps=-1,
                                                  lt_amd_accumulate_grad_batches=4)) == 250
    assert integration.total_optimizer_steps(host(estimated_stepping_batches=1000, lt_amd_accumulate_grad_batches=1)) == 1000


def test_unsupported_precision_and_accumulation_raise():
    from lightly_train_amd import integration

    H.install()
    kw = dict(arch="_vit_test", patch_size=14, img_size=224, global_batch_size=16, total_steps=2, seed=0)
    amd = H.build_reference_method(method_cls=integration.dinov2_amd_method_cls(), method_cls_kwargs=dict(device=torch.device("cpu")), **kw)
    amd.trainer.precision = "32-true"
    with pytest.raises(ValueError, match="bf16-mixed"):
        amd.impl()
    amd.trainer.precision = "bf16-mixed"
    with ops_emu.emulate(ops):
        assert amd.impl() is not None
    d = _build_dino(integration.dino_amd_method_cls())
    d.gradient_accumulation_steps = 2
    with ops_emu.emulate(ops), pytest.raises(NotImplementedError, match="accumulation"):
        d.training_step_impl({"views": [torch.zeros(8, 3, 96, 96)] * 2, "filename": []}, 0)


def test_dino_binding_converts_the_deprecated_epoch_arguments():
    """`warmup_teacher_temp_epochs` / `student_freeze_last_layer_epochs` (resolve_auto leaves the *_steps fields at None): converted with the
    trainer's steps per epoch as dino.py:301-312 / :450-468 do, not replaced by the step defaults."""
    from lightly_train_amd import integration

    d = _build_dino(integration.dino_amd_method_cls())
    a = d.method_args
    a.warmup_teacher_temp_steps, a.warmup_teacher_temp_epochs = None, 3
    a.student_freeze_last_layer_steps, a.student_freeze_last_layer_epochs = None, 1
    d.trainer.max_epochs, d.trainer.estimated_stepping_batches = 4, 20         # 5 steps per epoch
    with ops_emu.emulate(ops):
        ia = d.impl().method_args
    assert ia.student_freeze_last_layer_steps == 5
    assert ia.warmup_teacher_temp_steps == min(15, int(20 * a.warmup_teacher_temp_max_steps_fraction))


def test_dinov31_binding_three_steps_equal_the_reference_class():
    """`DINOv31AMD(Method)` (integration.dinov31_amd_method_cls) behind the reference's own DINOv31 constructor: the three steps of the
    fixture recipe (step 0 without PaKA, `paka_start_step = 1`) driven through the binding's hook equal the reference class driven in
    Lightning's hook order -- loss terms incl. `paka_loss`, and every student / EMA-teacher tensor incl. both PaKA heads afterwards."""
    from lightly_train_amd import integration
    from oracle import make_dinov31_fixture as MK

    H.install()
    from lightly_train._methods.dinov2.dinov2 import DINOv2AdamWViTArgs
    from lightly_train._methods.dinov31.dinov31 import DINOv31, DINOv31Args
    from lightly_train._models.dinov2_vit.dinov2_vit import DINOv2ViTModelWrapper
    from lightly_train._models.dinov2_vit.dinov2_vit_src.models import vision_transformer as vits
    from lightly_train._models.embedding_model import EmbeddingModel

    def make(cls, **kw):
        torch.manual_seed(11); random.seed(11)
        model = vits.DinoVisionTransformer(img_size=MK.G_SIZE, patch_size=16, embed_dim=64, depth=2, num_heads=1, mlp_ratio=4.0, init_values=1.0,
                                           drop_path_rate=0.0, ffn_layer="mlp", block_chunks=0, interpolate_offset=0.1)
        wrapped = DINOv2ViTModelWrapper(model)
        margs = DINOv31Args(output_dim=512, hidden_dim=128, dino_bottleneck_dim=64, koleo_loss_weight=0.0, paka_num_local=MK.K_PAKA, paka_start_step=1, paka_weight=0.7)
        oargs = DINOv2AdamWViTArgs()
        margs.resolve_auto(scaling_info=None, optimizer_args=oargs, wrapped_model=wrapped)
        m = cls(method_args=margs, optimizer_args=oargs, embedding_model=EmbeddingModel(wrapped_model=wrapped), global_batch_size=MK.B, num_input_channels=3, **kw)
        m.trainer = H.MockTrainer(20)
        return m

    ref = make(DINOv31)
    cls = integration.dinov31_amd_method_cls()
    amd = make(cls, device=torch.device("cpu"))
    assert list(amd.state_dict()) == list(ref.state_dict())
    runner = H.ReferenceRunner(ref)
    with ops_emu.emulate(ops):
        exactify(amd.impl())
        for step in range(3):
            views, geoms = MK.synth_batch(8100 + step)
            batch = {"views": [v.clone() for v in views], "filename": [], "geometries": geoms}
            random.seed(410 + step); torch.manual_seed(410 + step)
            res = ref.training_step_impl(batch, step)
            res.loss.backward()
            ref.on_before_optimizer_step(runner.optim)
            torch.nn.utils.clip_grad_norm_([p for g in runner.optim.param_groups for p in g["params"]], ref.method_args.gradient_clip_val)
            runner.optim.step(); runner.optim.zero_grad(set_to_none=True); runner.sched.step()
            ref.trainer.global_step += 1
            try:
                ref.on_train_batch_end(None, batch, step)
            except Exception:
                pass
            want = {k.split("/")[-1]: float(v) for k, v in res.log_dict.items()}
            random.seed(410 + step); torch.manual_seed(410 + step)
            amd.trainer.global_step = step
            got_res = amd.training_step_impl({"views": views, "filename": [], "geometries": geoms}, step)
            amd.trainer.global_step = step + 1
            got = {k.split("/")[-1]: float(v) for k, v in got_res.log_dict.items()}
            assert set(got) == set(want)
            for k in ("dino_global_loss", "dino_local_loss", "ibot_loss") + (("paka_loss",) if step >= 1 else ()):
                assert got[k] == pytest.approx(want[k], rel=5e-5, abs=5e-5), (step, k)
        sd, rsd = amd.state_dict(), ref.state_dict()
        for k in rsd:
            if k.endswith("paka_head.4.bias"):
                continue      # no gradient (a bias in front of the centring): AdamW steps on summation round-off of opposite signs
            assert torch.allclose(sd[k].float(), rsd[k].float(), atol=5e-5), (k, (sd[k].float() - rsd[k].float()).abs().max().item())


def _build_dino(cls, seed=0, backbone="vit"):
    """The reference's DINO constructor calls (as in oracle/make_golden.py::make_dino_v1) around `cls`."""
    H.install()
    from lightly_train._methods.dino.dino import DINO, DINOArgs
    from lightly_train._models.dinov2_vit.dinov2_vit import DINOv2ViTModelWrapper
    from lightly_train._models.dinov2_vit.dinov2_vit_src.models import vision_transformer as v2
    from lightly_train._models.embedding_model import EmbeddingModel
    from lightly_train._scaling import ScalingInfo

    torch.manual_seed(seed)
    if backbone == "resnet":
        from lightly_train._models.torchvision.resnet import ResNetModelWrapper
        from oracle import resnet_oracle as OR

        model = OR.ResNet((1, 1, 1, 1), width=8)
        with torch.no_grad():
            for n_, prm in model.named_parameters():
                if "bn" in n_ or "downsample.1" in n_:
                    prm.add_(0.2 * torch.randn_like(prm))
        wrapped = ResNetModelWrapper(model)
    else:
        model = v2.DinoVisionTransformer(img_size=96, patch_size=16, embed_dim=64, depth=2, num_heads=1, mlp_ratio=4.0, init_values=0.1, drop_path_rate=0.0,
                                         ffn_layer="mlp", block_chunks=0, interpolate_offset=0.1)
        wrapped = DINOv2ViTModelWrapper(model)
    margs = DINOArgs(hidden_dim=128, bottleneck_dim=64, output_dim=512, student_freeze_last_layer_steps=1, teacher_temp=0.07, warmup_teacher_temp=0.04,
                     warmup_teacher_temp_steps=3, momentum_start=0.99)
    oargs = DINO.optimizer_args_cls("auto")()
    margs.resolve_auto(scaling_info=ScalingInfo(dataset_size=1000, epochs=1), optimizer_args=oargs, wrapped_model=wrapped)
    kw = dict(device=torch.device("cpu")) if cls is not DINO else {}
    m = cls(method_args=margs, optimizer_args=oargs, embedding_model=EmbeddingModel(wrapped_model=wrapped), global_batch_size=8, num_input_channels=3, **kw)
    m.trainer = H.MockTrainer(20)
    m.current_epoch = 0
    return m


def _exactify_conv(m):
    m.ws = F32Workspace(torch.device("cpu"))
    for fp in (m.student, m.teacher):
        fp.bf16 = fp.data.clone()
        fp.b = {n: fp.bf16[fp.offsets[n]:fp.offsets[n] + fp.p[n].numel()].view(fp.shapes[n]) for n in fp.names}
    for e in (m.s_net, m.t_net):
        e.act_dtype = torch.float32
        e.w_stem = e.w_stem.float()
    for h in (m.s_head, m.t_head):
        h.wn = h.wn.float()
    m._refresh_derived()


@pytest.mark.parametrize("backbone", ["vit", "resnet"])
def test_dino_v1_binding_two_steps_equal_the_reference_class(backbone):
    """`DINOAMD(Method)` (integration.dino_amd_method_cls) behind the reference's own DINO constructor: two steps with the method's "auto"
    optimizer (SGD), across the last-layer unfreeze, equal the reference class driven through Lightning's hook order -- loss and every
    student / teacher tensor; state_dict keys in the reference's order."""
    H.install()
    from lightly_train._methods.dino.dino import DINO
    from lightly_train_amd import integration

    ref = _build_dino(DINO, backbone=backbone)
    amd = _build_dino(integration.dino_amd_method_cls(), backbone=backbone)
    assert list(amd.state_dict()) == list(ref.state_dict())
    [opt], [sched] = ref.configure_optimizers()
    sched = sched["scheduler"]
    g = torch.Generator().manual_seed(3)
    with ops_emu.emulate(ops):
        m = amd.impl()
        (_exactify_conv if backbone == "resnet" else exactify)(m)
        for step in range(2):
            views = [torch.randn(8, 3, 96, 96, generator=g) for _ in range(2)] + [torch.randn(8, 3, 48, 48, generator=g) for _ in range(2)]
            res = ref.training_step_impl({"views": [v.clone() for v in views], "filename": []}, 0)
            res.loss.backward()
            ref.on_before_optimizer_step(opt)
            torch.nn.utils.clip_grad_norm_([p for g_ in opt.param_groups for p in g_["params"] if p.grad is not None], 3.0)
            opt.step(); opt.zero_grad(set_to_none=True); sched.step()
            ref.trainer.global_step += 1
            got = drive(amd, views, step)
            assert got["loss"] == pytest.approx(float(res.loss), rel=3e-5), step
        sd, rsd = amd.state_dict(), ref.state_dict()
        assert list(sd) == list(rsd)
        for k in rsd:
            assert torch.allclose(sd[k].float(), rsd[k].float(), atol=3e-5), (k, (sd[k].float() - rsd[k].float()).abs().max().item())
        assert type(m).__name__ == ("DINOResNet" if backbone == "resnet" else "DINO")
    assert integration.install_as("dino") is integration.dino_amd_method_cls()


def _build_dv3(cls, s_kind, seed=11):
    """The reference's DistillationV3 constructor calls (oracle/make_golden.py::make_distill_case) around `cls`: frozen DINOv3 ViT teacher
    (D = 64), student of kind `s_kind`."""
    H.install()
    from lightly_train._methods.distillationv3.distillationv3 import DistillationV3AdamWArgs, DistillationV3Args
    from lightly_train._models.dinov2_vit.dinov2_vit import DINOv2ViTModelWrapper
    from lightly_train._models.dinov2_vit.dinov2_vit_src.models import vision_transformer as v2
    from lightly_train._models.dinov3.dinov3_src.models import vision_transformer as v3
    from lightly_train._models.dinov3.dinov3_vit import DINOv3ViTModelWrapper
    from lightly_train._models.embedding_model import EmbeddingModel

    torch.manual_seed(seed)
    t = v3.DinoVisionTransformer(img_size=64, patch_size=16, embed_dim=64, depth=2, num_heads=1, ffn_ratio=4.0, qkv_bias=True, layerscale_init=0.5,
                                 norm_layer="layernormbf16", ffn_layer="mlp", n_storage_tokens=4, mask_k_bias=True, pos_embed_rope_base=100.0,
                                 pos_embed_rope_dtype="fp32", pos_embed_rope_rescale_coords=2)
    t.init_weights()
    if s_kind == "resnet":
        from lightly_train._models.torchvision.resnet import ResNetModelWrapper
        from oracle import resnet_oracle as OR

        s_model = OR.ResNet((1, 1, 1, 1), width=8)
        with torch.no_grad():
            for n_, prm in s_model.named_parameters():
                if "bn" in n_ or "downsample.1" in n_:
                    prm.add_(0.2 * torch.randn_like(prm))
        sw = ResNetModelWrapper(s_model)
    elif s_kind == "dinov3":
        s_model = v3.DinoVisionTransformer(img_size=64, patch_size=16, embed_dim=64, depth=2, num_heads=1, ffn_ratio=4.0, qkv_bias=True, layerscale_init=0.1,
                                           norm_layer="layernormbf16", ffn_layer="mlp", n_storage_tokens=4, mask_k_bias=True, pos_embed_rope_base=100.0,
                                           pos_embed_rope_dtype="fp32", pos_embed_rope_rescale_coords=2)
        s_model.init_weights()
        sw = DINOv3ViTModelWrapper(s_model)
    else:
        s_model = v2.DinoVisionTransformer(img_size=64, patch_size=16, embed_dim=64, depth=2, num_heads=1, mlp_ratio=4.0, init_values=0.1, drop_path_rate=0.0,
                                           ffn_layer="mlp", block_chunks=0, interpolate_offset=0.1)
        sw = DINOv2ViTModelWrapper(s_model)
    margs = DistillationV3Args(queue_size=32, teacher=DINOv3ViTModelWrapper(t))
    oargs = DistillationV3AdamWArgs()
    oargs.resolve_auto(wrapped_model=sw)
    kw = dict(device=torch.device("cpu")) if cls.__name__.endswith("AMD") else {}
    m = cls(method_args=margs, optimizer_args=oargs, embedding_model=EmbeddingModel(wrapped_model=sw), global_batch_size=8, num_input_channels=3, **kw)
    m.trainer = H.MockTrainer(20)
    return m


@pytest.mark.parametrize("s_kind", ["dinov2", "dinov3", "resnet"])
def test_distillationv3_binding_two_steps_equal_the_reference_class(s_kind):
    """`DistillationV3AMD(Method)` (BASELINE configs[3]'s method) behind the reference's own constructor, for the three student families:
    two steps with identical mixup draws equal the reference class driven through Lightning's hook order -- loss terms, student, heads and
    queue; the teacher's keys are absent from a saved checkpoint and restored on load, as in the reference."""
    H.install()
    from lightly_train._methods.distillationv3.distillationv3 import DistillationV3
    from lightly_train_amd import integration
    from test_distillation_methods_cpu import exactify as exactify_distill

    ref = _build_dv3(DistillationV3, s_kind)
    amd = _build_dv3(integration.distillationv3_amd_method_cls(), s_kind)
    assert list(amd.state_dict()) == list(ref.state_dict())
    [opt], [sched] = ref.configure_optimizers()
    sched = sched["scheduler"]
    g = torch.Generator().manual_seed(9)
    with ops_emu.emulate(ops):
        exactify_distill(amd.impl())
        for step in range(2):
            x = torch.randn(8, 3, 64, 64, generator=g)
            torch.manual_seed(600 + step)
            res = ref.training_step_impl({"views": [x.clone()], "filename": []}, 0)
            res.loss.backward()
            torch.nn.utils.clip_grad_norm_([p for g_ in opt.param_groups for p in g_["params"] if p.grad is not None], 1.0)
            opt.step(); opt.zero_grad(set_to_none=True); sched.step()
            ref.trainer.global_step += 1
            torch.manual_seed(600 + step)
            batch = {"views": [x], "filename": []}
            amd.trainer.global_step = step
            got = amd.training_step_impl(batch, step)
            amd.trainer.global_step = step + 1
            assert float(got.loss) == pytest.approx(float(res.loss), rel=5e-5), (s_kind, step)
            for k in ("train_loss/local_loss", "train_loss/global_loss"):
                assert float(got.log_dict[k]) == pytest.approx(float(res.log_dict[k]), rel=5e-5, abs=1e-6), (s_kind, step, k)
        sd, rsd = amd.state_dict(), ref.state_dict()
        assert list(sd) == list(rsd)
        for k in rsd:
            if k.endswith("num_batches_tracked"):
                assert int(sd[k]) == int(rsd[k]), k
            else:
                assert torch.allclose(sd[k].float(), rsd[k].float(), atol=3e-5), (k, (sd[k].float() - rsd[k].float()).abs().max().item())
        ckpt = {"state_dict": dict(sd)}
        amd.on_save_checkpoint(ckpt)
        assert not any(k.startswith("teacher_embedding_model.") for k in ckpt["state_dict"]) and "amd_optimizer_state" in ckpt
        amd2 = _build_dv3(integration.distillationv3_amd_method_cls(), s_kind, seed=12)
        amd2.on_load_checkpoint(ckpt)
        assert any(k.startswith("teacher_embedding_model.") for k in ckpt["state_dict"])
        exactify_distill(amd2.impl())
        assert torch.equal(amd2.impl().student.data, amd.impl().student.data) and amd2.impl().opt_step == 2
    assert integration.install_as("distillation") is integration.distillationv3_amd_method_cls()


def _build_d12(kind, amd, seed=21, optimizer="auto"):
    """The reference's Distillation / DistillationV2 constructor calls (oracle/make_golden.py::make_distill12) around the reference class
    or its MI355X binding; `get_teacher` (a registry lookup by model NAME) is replaced by a function returning a locally built teacher."""
    import importlib

    H.install()
    from lightly_train._models.dinov2_vit.dinov2_vit import DINOv2ViTModelWrapper
    from lightly_train._models.dinov2_vit.dinov2_vit_src.models import vision_transformer as v2
    from lightly_train._models.embedding_model import EmbeddingModel
    from lightly_train_amd import integration

    torch.manual_seed(seed)
    t = v2.DinoVisionTransformer(img_size=112, patch_size=14, embed_dim=64, depth=3, num_heads=1, mlp_ratio=4.0, init_values=0.5, drop_path_rate=0.0,
                                 ffn_layer="mlp", block_chunks=0, interpolate_offset=0.1)
    t.eval()
    for prm in t.parameters():
        prm.requires_grad_(False)
    s_model = v2.DinoVisionTransformer(img_size=112, patch_size=16, embed_dim=64, depth=2, num_heads=1, mlp_ratio=4.0, init_values=0.1, drop_path_rate=0.0,
                                       ffn_layer="mlp", block_chunks=0, interpolate_offset=0.1)
    sw = DINOv2ViTModelWrapper(s_model)
    if kind == "v1":
        mod = importlib.import_module("lightly_train._methods.distillation.distillation")
        margs, ref_cls = mod.DistillationArgs(queue_size=32, teacher="local"), mod.Distillation
    else:
        mod = importlib.import_module("lightly_train._methods.distillationv2.distillationv2")
        margs, ref_cls = mod.DistillationV2Args(teacher="local"), mod.DistillationV2
    oargs = ref_cls.optimizer_args_cls(optimizer)()          # "auto" = the method's LARS arguments
    mod.get_teacher = lambda *a, **k: t
    cls = integration.distillation12_amd_method_cls(kind) if amd else ref_cls
    kw = dict(device=torch.device("cpu")) if amd else {}
    m = cls(method_args=margs, optimizer_args=oargs, embedding_model=EmbeddingModel(wrapped_model=sw), global_batch_size=8, num_input_channels=3, **kw)
    m.trainer = H.MockTrainer(20)
    return m


@pytest.mark.parametrize("kind", ["v1", "v2"])
def test_distillation_v1_v2_bindings_two_steps_equal_the_reference_class(kind):
    """`DistillationAMD` / `DistillationV2AMD` with the methods' "auto" optimizer (LARS, around the restated lightly.utils.lars.LARS):
    two steps equal the reference classes -- loss, student, head, queue."""
    from test_distillation_methods_cpu import exactify as exactify_distill

    ref, amd = _build_d12(kind, False), _build_d12(kind, True)
    assert list(amd.state_dict()) == list(ref.state_dict()) and type(amd).__name__ == ("DistillationAMD" if kind == "v1" else "DistillationV2AMD")
    [opt], [sched] = ref.configure_optimizers()
    sched = sched["scheduler"]
    g = torch.Generator().manual_seed(13)
    with ops_emu.emulate(ops):
        exactify_distill(amd.impl())
        assert amd.impl().optimizer == "lars"
        for step in range(2):
            x = torch.randn(8, 3, 112, 112, generator=g)
            torch.manual_seed(700 + step)
            res = ref.training_step_impl({"views": [x.clone()], "filename": []}, 0)
            res.loss.backward()
            torch.nn.utils.clip_grad_norm_([p for g_ in opt.param_groups for p in g_["params"] if p.grad is not None], 1.0)
            opt.step(); opt.zero_grad(set_to_none=True); sched.step()
            ref.trainer.global_step += 1
            torch.manual_seed(700 + step)
            amd.trainer.global_step = step
            got = amd.training_step_impl({"views": [x], "filename": []}, step)
            amd.trainer.global_step = step + 1
            assert float(got.loss) == pytest.approx(float(res.loss), rel=5e-5), (kind, step)
        sd, rsd = amd.state_dict(), ref.state_dict()
        assert list(sd) == list(rsd)
        for k in rsd:
            assert torch.allclose(sd[k].float(), rsd[k].float(), atol=5e-5), (k, (sd[k].float() - rsd[k].float()).abs().max().item())
        ckpt = {"state_dict": dict(sd)}
        amd.on_save_checkpoint(ckpt)
        assert not any(k.startswith("teacher_embedding_model.") for k in ckpt["state_dict"])
