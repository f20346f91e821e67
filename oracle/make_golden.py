"""TEST INFRASTRUCTURE ONLY.  Generates tests/golden/*.pt by running the REFERENCE's own
code (imported from /root/reference through oracle/ref_harness.py) on CPU in fp32.

Run in the build container:  python -m oracle.make_golden
The fixtures are committed; the GPU box (no /root/reference) only reads them.

Every fixture stores: the initial student/teacher state (reference key names), the RNG
seeds the synthetic views are regenerated from (+ a checksum), the masks the reference
sampled, and the reference's outputs: head logits captured by forward hooks, the four
loss terms, total loss, grad-norm, and student/teacher parameters + loss centers after
each optimizer step (Lightning hook order, see ReferenceRunner).
"""
from __future__ import annotations

import math
import os
import random
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import dinov2_oracle as O  # noqa: E402
from oracle import ref_harness as H  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def synth_views(seed: int, b: int, g_size: int, l_size: int, n_local: int):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(b, 3, g_size, g_size, generator=g) for _ in range(2)] + [
        torch.randn(b, 3, l_size, l_size, generator=g) for _ in range(n_local)
    ]


def make_step_fixture(name: str, arch: str, model_kwargs: dict, method_kwargs: dict, cfg: dict, b: int,
                      g_size: int, l_size: int, n_local: int, n_steps: int, total_steps: int,
                      keep_params_every_step: bool = True, teacher_head_training: bool = False,
                      twin_tol_later_steps: float = 2e-5) -> None:
    H.install()
    from lightly_train._methods.dinov2 import utils as ref_utils

    m = H.build_reference_method(arch=arch, patch_size=cfg["patch_size"], img_size=g_size,
                                 model_kwargs=model_kwargs, method_kwargs=method_kwargs,
                                 global_batch_size=b, total_steps=total_steps, seed=1234)
    if teacher_head_training:   # what a Trainer that calls module.train() at fit start leaves behind (only BatchNorm heads care)
        m.teacher_head.train()
    r = H.ReferenceRunner(m)
    init = r.split_state()
    # teacher backbone == deepcopy(student backbone) at init (dinov2.py:196-204): store once.
    # The two heads are built independently (dinov2.py:215-241): store both.
    for k, v in init["teacher_backbone"].items():
        assert torch.equal(v, init["student_backbone"][k])
    init_small = {"student_backbone": init["student_backbone"], "student_head": init["student_head"],
                  "teacher_head": init["teacher_head"]}
    separate = bool(method_kwargs.get("ibot_separate_head", False))
    if separate:
        init_small["student_ibot_head"] = init["student_ibot_head"]
        init_small["teacher_ibot_head"] = init["teacher_ibot_head"]
    fixture = {"name": name, "cfg": cfg, "method_kwargs": method_kwargs, "b": b, "g_size": g_size, "l_size": l_size,
               "n_local": n_local, "total_steps": total_steps, "init": init_small, "steps": [],
               "teacher_head_training": teacher_head_training}

    # capture head outputs in call order: teacher(cls, patch), student(cls, patch, local)
    cap: dict = {}
    t_calls, s_calls = [], []
    def spy_forward(mod, sink):
        orig = mod.forward

        def fwd(x):
            out = orig(x)
            sink.append(out.detach().clone())
            return out

        mod.forward = fwd

    spy_forward(m.teacher_head.dino_head, t_calls)
    spy_forward(m.student_head.dino_head, s_calls)
    ti_calls, si_calls = [], []
    if separate:
        spy_forward(m.teacher_head.ibot_head, ti_calls)
        spy_forward(m.student_head.ibot_head, si_calls)
    # capture masks the reference sampled
    orig_ccm = ref_utils.create_collated_masks
    import lightly_train._methods.dinov2.dinov2 as ref_dinov2

    def spy_ccm(**kw):
        out = orig_ccm(**kw)
        cap["masks"] = {k: v.clone() for k, v in out.items()}
        return out

    ref_dinov2.create_collated_masks = spy_ccm

    # oracle twin (must agree bit-for-bit on the first steps)
    o = O.OracleDINOv2(init["student_backbone"], init["student_head"], cfg,
                       args=dict(output_dim=method_kwargs.get("output_dim", 65536),
                                 hidden_dim=method_kwargs.get("hidden_dim", 2048),
                                 bottleneck_dim=method_kwargs.get("dino_bottleneck_dim", 256),
                                 center_method=method_kwargs.get("center_method", "softmax"),
                                 teacher_head_training=teacher_head_training),
                       global_batch_size=b, total_steps=total_steps,
                       teacher_backbone=init["teacher_backbone"], teacher_head=init["teacher_head"],
                       student_ibot_head=init["student_ibot_head"] if separate else None,
                       teacher_ibot_head=init["teacher_ibot_head"] if separate else None)

    for step in range(n_steps):
        views = synth_views(1000 + step, b, g_size, l_size, n_local)
        t_calls.clear(); s_calls.clear(); ti_calls.clear(); si_calls.clear()
        random.seed(77 + step)
        logs = r.train_step(views)
        random.seed(77 + step)
        ologs = o.train_step(views)
        tol = 2e-5 if step == 0 else twin_tol_later_steps
        for k in ("loss", "dino_global_loss", "dino_local_loss", "ibot_loss", "koleo_loss"):
            if k == "koleo_loss" and tol > 2e-5:
                continue
            assert abs(logs[k] - ologs[k]) <= tol * max(1.0, abs(logs[k])), (name, step, k, logs[k], ologs[k])
        rec = {
            "view_seed": 1000 + step,
            "view_checksum": float(sum(v.double().sum() for v in views)),
            "masks": cap["masks"],
            "logs": logs,
            "teacher_cls_logits": t_calls[0], "teacher_patch_logits": ti_calls[0] if separate else t_calls[1],
            "student_cls_logits": s_calls[0], "student_patch_logits": si_calls[0] if separate else s_calls[1],
            "student_local_logits": (s_calls[1] if separate else s_calls[2]) if n_local > 0 else None,
            "dino_center": m.dino_loss.center.detach().clone(),
            "ibot_center": m.ibot_loss.center.detach().clone(),
        }
        if keep_params_every_step or step == n_steps - 1:
            rec["state"] = r.split_state()
        fixture["steps"].append(rec)
        print(name, step, {k: round(v, 6) for k, v in logs.items()})
    ref_dinov2.create_collated_masks = orig_ccm
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".pt")
    torch.save(fixture, path)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def make_loss_kats() -> None:
    """Literal tensors of the reference's own known-answer tests, evaluated by the reference
    code (tests/_methods/dinov2/test_dinov2_loss.py:84-103,179-211; tests/test__torch_helpers.py:38-72)."""
    H.install()
    from lightly_train._methods.dinov2.dinov2_loss import DINOLoss, IBOTPatchLoss

    dl = DINOLoss(out_dim=2, student_temp=0.1, center_momentum=0.9)
    t = torch.tensor([[0.1, 0.2], [0.3, 0.4], [0.5, 0.6]])
    s = torch.tensor([[0.7, 0.8], [0.9, 1.0], [1.1, 1.2]])
    tp = dl.softmax_center_teacher(t, teacher_temp=0.04)
    dino = float(dl.forward([s, s], [tp, tp]))
    assert abs(dino - 1.5565) < 1.5565e-4, dino
    il = IBOTPatchLoss(patch_out_dim=2, student_temp=0.2, center_momentum=0.9)
    mask = torch.tensor([[True, False, True, False], [False, False, False, True], [False, False, False, False]])
    tpc = il.softmax_center_teacher(t.unsqueeze(0), teacher_temp=0.1)
    ibot = float(il.forward_masked(teacher_patch_tokens_masked=tpc, student_patch_tokens_masked=s,
                                   student_masks_flat=mask))
    assert abs(ibot - 0.4057) < 0.4057e-4, ibot
    sp, tpp = s, t
    # random medium-size cases for sinkhorn / softmax / CE
    g = torch.Generator().manual_seed(3)
    logits = torch.randn(24, 384, generator=g)
    dl2 = DINOLoss(out_dim=384)
    sk = dl2.sinkhorn_knopp_teacher(logits.clone(), teacher_temp=0.05)
    il2 = IBOTPatchLoss(patch_out_dim=384)
    sk_ibot = il2.sinkhorn_knopp_teacher(logits.clone(), teacher_temp=0.05,
                                         n_masked_patches_tensor=torch.tensor([24], dtype=torch.long))
    dl2.center = torch.randn(1, 384, generator=g) * 0.1
    sm = dl2.softmax_center_teacher(logits, teacher_temp=0.05)
    s2 = torch.randn(24, 384, generator=g)
    ce = float(dl2.forward(s2.chunk(2), list(sm.view(2, 12, 384))))
    out = {"dino_kat": dino, "ibot_kat": ibot, "s": s, "t": t, "sp": sp, "tpp": tpp, "mask": mask,
           "logits": logits, "sinkhorn": sk, "sinkhorn_ibot": sk_ibot, "center": dl2.center.clone(),
           "softmax_center": sm, "student": s2, "dino_ce_2x2": ce}
    torch.save(out, os.path.join(OUT, "loss_kats.pt"))
    print("loss KATs", dino, ibot, ce)


def main() -> None:
    os.makedirs(OUT, exist_ok=True)
    if "--reg4-only" in sys.argv:
        make_reg4_swiglu()
        return
    if "--bn-only" in sys.argv:
        make_bn_heads()
        return
    if "--dinov3-only" in sys.argv:
        make_dinov3_vit()
        return
    if "--dino-v1-only" in sys.argv:
        make_dino_v1_kats()
        make_dino_v1("sgd")
        make_dino_v1("adamw")
        make_dino_v1("sgd", backbone="resnet")
        return
    if "--lars-only" in sys.argv:
        make_distill12("v1", optimizer="lars")
        return
    if "--distill-v2-mlp-only" in sys.argv:
        make_distill12("v2", n_layers=3)
        return
    if "--distill12-only" in sys.argv:
        make_distill12("v1")
        make_distill12("v2")
        return
    if "--distill-only" in sys.argv:
        make_distill()
        return
    make_loss_kats()
    small_head = dict(output_dim=512, hidden_dim=64, dino_bottleneck_dim=32)
    # (a) the reference tests' own toy model: D=8, depth 3, 2 heads (head_dim 4)
    make_step_fixture("step_vittest_softmax", "_vit_test", {}, dict(small_head),
                      dict(patch_size=16, num_heads=2, depth=3), b=8, g_size=64, l_size=32, n_local=4,
                      n_steps=3, total_steps=20)
    make_step_fixture("step_vittest_sinkhorn", "_vit_test", {}, dict(small_head, center_method="sinkhorn_knopp"),
                      dict(patch_size=16, num_heads=2, depth=3), b=8, g_size=64, l_size=32, n_local=4,
                      n_steps=2, total_steps=20)
    make_step_fixture("step_vittest_sephead", "_vit_test", {}, dict(small_head, ibot_separate_head=True),
                      dict(patch_size=16, num_heads=2, depth=3), b=8, g_size=64, l_size=32, n_local=2,
                      n_steps=1, total_steps=20, keep_params_every_step=False)
    # (b) head_dim 64 (the MFMA attention path): D=64, depth 2, 1 head; 96->6x6 global, 48->3x3 local
    make_step_fixture("step_d64_softmax", "DinoVisionTransformer",
                      dict(embed_dim=64, depth=2, num_heads=1, mlp_ratio=4.0),
                      dict(output_dim=512, hidden_dim=128, dino_bottleneck_dim=64),
                      dict(patch_size=16, num_heads=1, depth=2), b=8, g_size=96, l_size=48, n_local=2,
                      n_steps=2, total_steps=50, keep_params_every_step=False)
    make_reg4_swiglu()
    make_dinov3_vit()
    make_distill()


def make_bn_heads() -> None:
    """batch_norm=True (dinov2_head.py:86-92): BatchNorm1d after the two hidden Linear layers of every head.  Two steps with the
    teacher heads as the reference constructs them (eval: running estimates), one step with them in train() (batch statistics).
    LayerScale starts at 1.0 here: at the usual 1e-5 the cls rows of a fresh model agree to ~1e-5 relative, BatchNorm divides their
    spread by sqrt(eps) and the step turns chaotic (the fp32 restatement and the reference then agree to 1e-7 on step 0 but only to
    4 % in the local-crop logits one step later, and a bf16 run normalises rounding noise) -- nothing a parity fixture can pin."""
    model = dict(embed_dim=64, depth=2, num_heads=1, mlp_ratio=4.0, init_values=1.0)
    head = dict(output_dim=512, hidden_dim=128, dino_bottleneck_dim=64, batch_norm=True)
    cfg = dict(patch_size=16, num_heads=1, depth=2)
    make_step_fixture("step_d64_bn", "DinoVisionTransformer", model, head, cfg, b=8, g_size=96, l_size=48, n_local=2,
                      n_steps=2, total_steps=50, keep_params_every_step=False)
    make_step_fixture("step_d64_bn_sephead_ttrain", "DinoVisionTransformer", model, dict(head, ibot_separate_head=True), cfg, b=8,
                      g_size=96, l_size=48, n_local=2, n_steps=1, total_steps=50, teacher_head_training=True)


def make_distill() -> None:
    """(e) DistillationV3 (config 4 path): the reference's own DistillationV3 class with a frozen DINOv3 ViT teacher (the
    dinov3_vitl16 recipe at D=64) and a DINOv2-ViT student (D=64, depth 2), queue 32, AdamW, 3 steps: 64^2 images with a /16
    student (equal grids), and 112^2 images with a /14 student (8x8 student grid resized bilinearly onto the 7x7 teacher grid)."""
    make_distill_case("distill_v3_d64", img=64, s_patch=16, b=8)
    make_distill_case("distill_v3_d64_p14", img=112, s_patch=14, b=4)
    make_distill_case("distill_v3_d64_v3s", img=64, s_patch=16, b=8, s_kind="dinov3")   # DINOv3 student (train-mode RoPE rescale)
    # convolutional student (BASELINE configs[3] names torchvision/resnet50): the reference's own ResNetModelWrapper + DistillationV3
    # around the restated torchvision ResNet (oracle/resnet_oracle.py), tiny bottleneck net (1,1,1,1) x width 8 -> 2x2 map of 256
    # channels at 64^2, resized bilinearly onto the teacher's 4x4 grid; weight decay "auto" -> 1e-6 (distillationv3.py:163-170)
    make_distill_case("distill_v3_resnet", img=64, s_patch=16, b=8, s_kind="resnet")


def make_distill12(kind: str, optimizer: str = "adamw", n_layers: int = 1) -> None:
    """(f) Distillation (v1: pooled feature vs a queue, KL) and DistillationV2 (patch features of the last 2 teacher blocks, MSE): the
    reference's own classes with a frozen DINOv2 ViT teacher (D = 64, /14: 4x4 tokens at 56^2 ... here 8x8 at 112^2) and a DINOv2 ViT
    student (/16: 7x7 tokens, resized onto the teacher grid in v2), AdamW (the reference's "auto" LARS lives in un-vendored LightlySSL),
    3 optimizer steps.  `get_teacher` (which resolves a model NAME through the package registry) is replaced by a function that
    returns the locally built teacher; everything else is the reference's code."""
    H.install()
    import importlib

    from lightly_train._models.dinov2_vit.dinov2_vit import DINOv2ViTModelWrapper
    from lightly_train._models.dinov2_vit.dinov2_vit_src.models import vision_transformer as v2
    from lightly_train._models.embedding_model import EmbeddingModel
    from lightly_train._optim.adamw_args import AdamWArgs
    from oracle import distill_oracle as OD

    img, b, total, qsz = 112, 8, 20, 32
    torch.manual_seed(977)
    t = v2.DinoVisionTransformer(img_size=img, patch_size=14, embed_dim=64, depth=3, num_heads=1, mlp_ratio=4.0, init_values=0.5, drop_path_rate=0.0,
                                 ffn_layer="mlp", block_chunks=0, interpolate_offset=0.1)
    for n_, prm in t.named_parameters():   # biases / LayerNorm affine are initialised to constants: randomise for a real test
        if n_.endswith(".bias") or "norm" in n_:
            prm.data.add_(0.1 * torch.randn_like(prm))
    t.eval()
    for prm in t.parameters():
        prm.requires_grad_(False)
    s_model = v2.DinoVisionTransformer(img_size=img, patch_size=16, embed_dim=64, depth=2, num_heads=1, mlp_ratio=4.0, init_values=0.1,
                                       drop_path_rate=0.0, ffn_layer="mlp", block_chunks=0, interpolate_offset=0.1)
    sw = DINOv2ViTModelWrapper(s_model)
    if kind == "v1":
        mod = importlib.import_module("lightly_train._methods.distillation.distillation")
        margs = mod.DistillationArgs(queue_size=qsz, teacher="local")
        # "lars": the method's "auto" optimizer arguments (lr 1.8, momentum 0.9, weight decay 1e-6) around the restated
        # lightly.utils.lars.LARS that ref_harness registers (oracle/lars_oracle.py: parity unpinned for the optimizer rule itself)
        oargs = mod.DistillationLARSArgs() if optimizer == "lars" else mod.DistillationAdamWArgs()
        cls = mod.Distillation
    else:
        mod = importlib.import_module("lightly_train._methods.distillationv2.distillationv2")
        # n_layers > 1: the Linear-LayerNorm-GELU projection stack of DistillationV2Head (distillationv2.py:116-152), hidden width 48
        margs = mod.DistillationV2Args(teacher="local", n_projection_layers=n_layers, projection_hidden_dim=48)
        oargs = AdamWArgs()
        cls = mod.DistillationV2
    mod.get_teacher = lambda *a, **k: t      # the registry lookup by name is outside this path
    m = cls(method_args=margs, optimizer_args=oargs, embedding_model=EmbeddingModel(wrapped_model=sw), global_batch_size=b, num_input_channels=3)
    m.trainer = H.MockTrainer(total)
    [opt], [sched] = m.configure_optimizers()
    sched = sched["scheduler"]
    head = m.student_projection_head
    init = {"student_backbone": {k: v.detach().clone() for k, v in s_model.state_dict().items()},
            "head": {k: v.detach().clone() for k, v in head.state_dict().items()}}
    teacher_state = {k: v.detach().clone() for k, v in t.state_dict().items()}
    scfg = dict(patch_size=16, num_heads=1, depth=2, img_size=img, embed_dim=64, init_values=0.1)
    tcfg = dict(patch_size=14, num_heads=1, depth=3, img_size=img, embed_dim=64, init_values=0.5)
    o = None
    if n_layers == 1:   # (the restated twin covers the single-Linear head; the deeper stacks are pinned by this fixture alone)
        o = OD.OracleDistillation12(kind, init["student_backbone"], scfg, teacher_state, tcfg, init["head"], qsz, b, total, lr=float(oargs.lr),
                                    weight_decay=float(oargs.weight_decay), optimizer=optimizer)
        assert (o.n_decay, o.n_no_decay) == tuple(len(g["params"]) for g in opt.param_groups), ((o.n_decay, o.n_no_decay), [len(g["params"]) for g in opt.param_groups])
    steps = []
    for step in range(3):
        x = torch.randn(b, 3, img, img, generator=torch.Generator().manual_seed(2100 + step))
        torch.manual_seed(400 + step)
        lam = torch.empty(1).uniform_(0.0, 1.0).item()
        index = torch.randperm(b)
        torch.manual_seed(400 + step)
        lr_now = opt.param_groups[0]["lr"]
        res = m.training_step_impl({"views": [x], "filename": []}, 0)
        res.loss.backward()
        gnorm = torch.nn.utils.clip_grad_norm_([p for g in opt.param_groups for p in g["params"]], 1.0)
        opt.step(); opt.zero_grad(set_to_none=True); sched.step()
        m.trainer.global_step += 1
        logs = {"loss": float(res.loss.detach()), "grad_norm": float(gnorm), "lr": lr_now}
        if o is not None:
            ol = o.train_step(x, lam, index)
            for k in ("loss", "grad_norm"):
                assert abs(ol[k] - logs[k]) <= 2e-5 * max(1.0, abs(logs[k])), (kind, step, k, ol[k], logs[k])
        steps.append({"x_seed": 2100 + step, "lam": lam, "index": index.clone(), "logs": logs})
        print("distill", kind, step, {k: round(v, 6) for k, v in logs.items()})
    final = {"student_backbone": {k: v.detach().clone() for k, v in s_model.state_dict().items()},
             "head": {k: v.detach().clone() for k, v in head.state_dict().items()}}
    if kind == "v1":
        final["queue"] = m.teacher_queue.detach().clone()
        assert (o.queue - final["queue"]).abs().max().item() < 1e-6
    if o is not None:
        for k, v in final["student_backbone"].items():
            assert (o.sb[k].detach() - v).abs().max().item() <= 2e-6 + 2e-5 * v.abs().max().item(), k
    name = "distill_" + kind + "_d64" + ("_lars" if optimizer == "lars" else "") + (f"_mlp{n_layers}" if n_layers > 1 else "")
    torch.save({"kind": kind, "optimizer": optimizer, "b": b, "img": img, "total_steps": total, "queue_size": qsz, "lr": float(oargs.lr), "weight_decay": float(oargs.weight_decay),
                "student_cfg": scfg, "teacher_cfg": tcfg, "teacher_state": teacher_state, "init": init, "steps": steps, "final": final,
                "n_projection_layers": n_layers, "projection_hidden_dim": 48,
                "state_dict_keys": list(m.state_dict().keys())}, os.path.join(OUT, name + ".pt"))
    print("wrote", name, os.path.getsize(os.path.join(OUT, name + ".pt")) // 1024, "KiB")


def make_dino_v1_kats() -> None:
    """Anchors for the restated LightlySSL pieces of DINO (oracle/dino_oracle.py), evaluated by the reference's VENDORED twins:
    `dinov2_loss.DINOLoss` (softmax-center teacher, the cross-entropy summed over view pairs, the center update) and
    `dinov2_head.DINOv2ProjectionHead` (same function as lightly's DINOProjectionHead under other attribute names)."""
    H.install()
    from lightly_train._methods.dinov2.dinov2_head import DINOv2ProjectionHead
    from lightly_train._methods.dinov2.dinov2_loss import DINOLoss as VendoredDINOLoss

    g = torch.Generator().manual_seed(515)
    B, K, n_views, temp = 6, 32, 4, 0.05
    teacher = [torch.randn(B, K, generator=g) for _ in range(2)]
    student = [torch.randn(B, K, generator=g) for _ in range(n_views)]
    center = 0.3 * torch.randn(1, K, generator=g)
    L = VendoredDINOLoss(out_dim=K, student_temp=0.1, center_momentum=0.9)
    L.center = center.clone()
    tp = [L.softmax_center_teacher(t, temp) for t in teacher]
    every_pair = L(student, tp)                                   # sum over all (s, t) of -mean_b sum_k t log_softmax(s / T_s)
    same_view = L([student[0]], [tp[0]]) + L([student[1]], [tp[1]])
    n_terms = 2 * n_views - 2
    L.update_center(torch.cat(teacher))
    L.apply_center_update()
    torch.manual_seed(99)
    head = DINOv2ProjectionHead(in_dim=16, out_dim=48, hidden_dim=24, bottleneck_dim=8)
    for prm in head.parameters():
        prm.data.add_(0.05 * torch.randn_like(prm))
    x = torch.randn(5, 16, generator=g)
    torch.save({"teacher": teacher, "student": student, "center": center, "teacher_temp": temp, "student_temp": 0.1, "center_momentum": 0.9,
                "loss": float((every_pair - same_view) / n_terms), "center_after": L.center.detach().clone(),
                "head_state": {k: v.detach().clone() for k, v in head.state_dict().items()}, "head_in": x, "head_out": head(x).detach().clone()},
               os.path.join(OUT, "dino_v1_kats.pt"))
    print("wrote dino_v1_kats")


def make_dino_v1(optimizer: str, backbone: str = "vit") -> None:
    """(g) DINO (LT/_methods/dino/dino.py:221-480): the reference's own `DINO` class around a DINOv2 ViT (D = 64, depth 2, /16; 96^2
    global and 48^2 local views, 2 + 2 views, batch 8), 4 optimizer steps with the last layer frozen during the first two
    (student_freeze_last_layer_steps=2) and the teacher temperature warming up over 3.  optimizer "sgd" = the method's "auto" arguments
    (DINOSGDArgs: lr 0.03, momentum 0.9, weight decay 1e-4), "adamw" = DINOAdamWArgs (lr 5e-4) with the weight decay scheduled 0.04 ->
    0.4.  The LightlySSL pieces (DINOLoss, DINOProjectionHead, get_weight_decay_parameters) are the restatements of oracle/dino_oracle.py
    that ref_harness registers -- parity unpinned for those; everything else that runs is the reference's code."""
    H.install()
    from lightly_train._methods.dino.dino import DINO, DINOAdamWArgs, DINOArgs, DINOSGDArgs
    from lightly_train._models.dinov2_vit.dinov2_vit import DINOv2ViTModelWrapper
    from lightly_train._models.dinov2_vit.dinov2_vit_src.models import vision_transformer as v2
    from lightly_train._models.embedding_model import EmbeddingModel
    from lightly_train._scaling import ScalingInfo

    b, g_size, l_size, n_local, total, n_steps = 8, 96, 48, 2, 50, 4

    def construct():
        torch.manual_seed(4242)
        if backbone == "resnet":
            # backbone "resnet": the reference's ResNetModelWrapper around the restated torchvision bottleneck ResNet (oracle/resnet_oracle.py:
            # layers (1, 1, 1, 1), width 8 -> 256 features), train-mode BatchNorm in student AND teacher (batch statistics per forward call,
            # running estimates moved by each); the EMA walks parameters only
            from lightly_train._models.torchvision.resnet import ResNetModelWrapper
            from oracle import resnet_oracle as OR

            model = OR.ResNet((1, 1, 1, 1), width=8)
            for n_, prm in model.named_parameters():
                if "bn" in n_ or "downsample.1" in n_:
                    prm.data.add_(0.2 * torch.randn_like(prm))
            wrapped = ResNetModelWrapper(model)
        else:
            model = v2.DinoVisionTransformer(img_size=g_size, patch_size=16, embed_dim=64, depth=2, num_heads=1, mlp_ratio=4.0, init_values=0.1,
                                             drop_path_rate=0.0, ffn_layer="mlp", block_chunks=0, interpolate_offset=0.1)
            for n_, prm in model.named_parameters():   # constants at init (biases 0, norm weights 1, mask token 0): randomise for a real test
                if n_.endswith(".bias") or "norm" in n_ or n_ == "mask_token":
                    prm.data.add_(0.1 * torch.randn_like(prm))
            wrapped = DINOv2ViTModelWrapper(model)
        margs = DINOArgs(hidden_dim=128, bottleneck_dim=64, output_dim=512, student_freeze_last_layer_steps=2, teacher_temp=0.07, warmup_teacher_temp=0.04,
                         warmup_teacher_temp_steps=3, momentum_start=0.99)
        if optimizer == "adamw":
            oargs = DINOAdamWArgs()
            margs.weight_decay_start, margs.weight_decay_end = 0.04, 0.4
        else:
            oargs = DINO.optimizer_args_cls("auto")()
            assert isinstance(oargs, DINOSGDArgs)
        margs.resolve_auto(scaling_info=ScalingInfo(dataset_size=1000, epochs=1), optimizer_args=oargs, wrapped_model=wrapped)
        m = DINO(method_args=margs, optimizer_args=oargs, embedding_model=EmbeddingModel(wrapped_model=wrapped), global_batch_size=b, num_input_channels=3)
        m.trainer = H.MockTrainer(total)
        m.current_epoch = 0
        return m, margs, oargs

    m, margs, oargs = construct()
    [opt], [sched] = m.configure_optimizers()
    sched = sched["scheduler"]
    groups = {g["name"]: len(g["params"]) for g in opt.param_groups}

    def split(sd):
        out = {"student_backbone": {}, "teacher_backbone": {}, "student_head": {}, "teacher_head": {}, "other": {}}
        for k, v in sd.items():
            for role in ("student", "teacher"):
                for pre, dst in ((f"{role}_embedding_model.wrapped_model._model.", f"{role}_backbone"),
                                 (f"{role}_embedding_model.wrapped_model._features.", f"{role}_backbone"), (f"{role}_projection_head.", f"{role}_head")):
                    if k.startswith(pre):
                        out[dst][k[len(pre):]] = v.detach().clone()
                        break
                else:
                    continue
                break
            else:
                out["other"][k] = v.detach().clone()
        return out

    init = split(m.state_dict())
    # yardstick for the bf16 kernels: the reference's own first step under torch.autocast("cpu", bfloat16) (= precision "bf16-mixed") from the
    # same state -- how far ITS mixed-precision gradients are from its fp32 ones, tensor by tensor
    my, _, _ = construct()     # (same seed, same constructor calls: the same initial state; deepcopy trips over nn.utils.weight_norm)
    with torch.autocast("cpu", dtype=torch.bfloat16):
        ry = my.training_step_impl({"views": synth_views(3000, b, g_size, l_size, n_local), "filename": []}, 0)
    ry.loss.float().backward()
    yard = {n: p.grad.detach().float().clone() for n, p in my.named_parameters() if p.grad is not None}
    yard_loss = float(ry.loss.detach())
    del my
    steps = []
    t_calls, s_calls = [], []
    for mod, sink in ((m.teacher_projection_head, t_calls), (m.student_projection_head, s_calls)):
        orig = mod.forward
        mod.forward = (lambda orig, sink: lambda x: (sink.append(orig(x)), sink[-1])[1])(orig, sink)
    for step in range(n_steps):
        views = synth_views(3000 + step, b, g_size, l_size, n_local)
        t_calls.clear(); s_calls.clear()
        res = m.training_step_impl({"views": views, "filename": []}, 0)
        res.loss.backward()
        m.on_before_optimizer_step(opt)
        hp = {g["name"]: {"lr": g["lr"], "weight_decay": g["weight_decay"]} for g in opt.param_groups}
        params = [p for g in opt.param_groups for p in g["params"] if p.grad is not None]
        gnorm = torch.nn.utils.clip_grad_norm_(params, 3.0)
        rec = {"view_seed": 3000 + step, "logs": {"loss": float(res.loss.detach()), "grad_norm": float(gnorm), **{k.split("/")[-1]: float(v) for k, v in res.log_dict.items()}},
               "hparams": hp, "teacher_logits": t_calls[0].detach().clone(), "student_global_logits": s_calls[0].detach().clone(),
               "student_local_logits": s_calls[1].detach().clone()}
        if step == 0:   # clipped gradients of the first step, by state_dict name
            names = {id(p): n for n, p in m.named_parameters()}
            rec["grads"] = {names[id(p)]: p.grad.detach().clone() for p in params}
            # (both sets of gradients unclipped for the comparison: clipping happened in place just above)
            cl = min(1.0, 3.0 / (float(gnorm) + 1e-6))
            errs = {k_: float((yard[k_] - g_ / cl).abs().max() / (g_ / cl).abs().max().clamp_min(1e-20)) for k_, g_ in rec["grads"].items() if k_ in yard}
            rec["bf16_autocast"] = {"loss": yard_loss, "grad_err": errs}
            ev = sorted(errs.values())
            print("  reference bf16 autocast vs its fp32: loss", round(yard_loss, 5), "grad err median", round(ev[len(ev) // 2], 4), "max", round(ev[-1], 4))
            rec["no_grad"] = sorted(names[id(p)] for g in opt.param_groups for p in g["params"] if p.grad is None)
        opt.step(); opt.zero_grad(set_to_none=True); sched.step()
        m.trainer.global_step += 1
        rec["center"] = m.criterion.center.value.detach().clone()
        steps.append(rec)
        print("dino_v1", optimizer, step, {k: round(v, 6) for k, v in rec["logs"].items()}, hp["params_last_layer"])
    final = split(m.state_dict())
    osd = opt.state_dict()
    name = ("dino_v1_resnet" if backbone == "resnet" else "dino_v1_d64") + ("_adamw" if optimizer == "adamw" else "")
    torch.save({"optimizer": optimizer, "backbone": backbone, "b": b, "g_size": g_size, "l_size": l_size, "n_local": n_local, "total_steps": total,
                "cfg": (dict(kind="resnet", layers=(1, 1, 1, 1), width=8) if backbone == "resnet" else
                        dict(patch_size=16, num_heads=1, depth=2, img_size=g_size, embed_dim=64, init_values=0.1)),
                "method_args": {k: getattr(margs, k) for k in ("hidden_dim", "bottleneck_dim", "output_dim", "student_freeze_last_layer_steps", "norm_last_layer",
                                                                 "teacher_temp", "warmup_teacher_temp", "warmup_teacher_temp_steps", "student_temp", "center_momentum",
                                                                 "momentum_start", "momentum_end", "weight_decay_start", "weight_decay_end", "warmup_steps",
                                                                 "warmup_max_steps_fraction", "lr_scale_method", "reference_batch_size")},
                "optimizer_args": oargs.model_dump(), "groups": groups, "init": init, "steps": steps, "final": final,
                "optimizer_state": osd, "state_dict_keys": list(m.state_dict().keys())}, os.path.join(OUT, name + ".pt"))
    print("wrote", name, os.path.getsize(os.path.join(OUT, name + ".pt")) // 1024, "KiB", groups)


def make_distill_case(name: str, img: int, s_patch: int, b: int, s_kind: str = "dinov2") -> None:
    H.install()
    from lightly_train._methods.distillationv3.distillationv3 import DistillationV3, DistillationV3AdamWArgs, DistillationV3Args
    from lightly_train._models.dinov2_vit.dinov2_vit import DINOv2ViTModelWrapper
    from lightly_train._models.dinov2_vit.dinov2_vit_src.models import vision_transformer as v2
    from lightly_train._models.dinov3.dinov3_src.models import vision_transformer as v3
    from lightly_train._models.dinov3.dinov3_vit import DINOv3ViTModelWrapper
    from lightly_train._models.embedding_model import EmbeddingModel
    from oracle import distill_oracle as OD

    torch.manual_seed(4321)
    t = v3.DinoVisionTransformer(img_size=img, patch_size=16, embed_dim=64, depth=2, num_heads=1, ffn_ratio=4.0, qkv_bias=True,
                                 layerscale_init=0.5, norm_layer="layernormbf16", ffn_layer="mlp", n_storage_tokens=4, mask_k_bias=True,
                                 pos_embed_rope_base=100.0, pos_embed_rope_dtype="fp32", pos_embed_rope_rescale_coords=2)
    t.init_weights()
    if s_kind == "resnet":
        from lightly_train._models.torchvision.resnet import ResNetModelWrapper
        from oracle import resnet_oracle as OR

        s_model = OR.ResNet((1, 1, 1, 1), width=8)
        for n_, prm in s_model.named_parameters():    # BatchNorm affine is initialised to constants: randomise for a real test
            if "bn" in n_ or "downsample.1" in n_:
                prm.data.add_(0.2 * torch.randn_like(prm))
        sw = ResNetModelWrapper(s_model)
    elif s_kind == "dinov3":
        s_model = v3.DinoVisionTransformer(img_size=img, patch_size=s_patch, embed_dim=64, depth=2, num_heads=1, ffn_ratio=4.0, qkv_bias=True,
                                           layerscale_init=0.1, norm_layer="layernormbf16", ffn_layer="mlp", n_storage_tokens=4,
                                           mask_k_bias=True, pos_embed_rope_base=100.0, pos_embed_rope_dtype="fp32",
                                           pos_embed_rope_rescale_coords=2)
        s_model.init_weights()
        sw = DINOv3ViTModelWrapper(s_model)
    else:
        s_model = v2.DinoVisionTransformer(img_size=img, patch_size=s_patch, embed_dim=64, depth=2, num_heads=1, mlp_ratio=4.0, init_values=0.1,
                                           drop_path_rate=0.0, ffn_layer="mlp", block_chunks=0, interpolate_offset=0.1)
        sw = DINOv2ViTModelWrapper(s_model)
    total, qsz = 20, 32
    margs = DistillationV3Args(queue_size=qsz, teacher=DINOv3ViTModelWrapper(t))
    oargs = DistillationV3AdamWArgs()
    oargs.resolve_auto(wrapped_model=sw)
    m = DistillationV3(method_args=margs, optimizer_args=oargs, embedding_model=EmbeddingModel(wrapped_model=sw), global_batch_size=b,
                       num_input_channels=3)
    m.trainer = H.MockTrainer(total)
    [opt], [sched] = m.configure_optimizers()
    sched = sched["scheduler"]
    teacher_state = {k: v.detach().clone() for k, v in t.state_dict().items()}
    init = {"student_backbone": {k: v.detach().clone() for k, v in s_model.state_dict().items()},
            "proj_global": {k: v.detach().clone() for k, v in m.student_projection_head_global.state_dict().items()},
            "proj_local": {k: v.detach().clone() for k, v in m.student_projection_head_local.state_dict().items()}}
    scfg = dict(patch_size=s_patch, num_heads=1, depth=2, img_size=img, embed_dim=64, init_values=0.1)
    if s_kind == "resnet":
        scfg = dict(kind="resnet", layers=(1, 1, 1, 1), width=8)
    if s_kind == "dinov3":
        scfg.update(rope_base=100.0, rope_rescale=2.0, ln_eps=1e-5, n_storage_tokens=4, kind="dinov3")
    tcfg = dict(patch_size=16, num_heads=1, depth=2, rope_base=100.0, ln_eps=1e-5, embed_dim=64, n_storage_tokens=4, img_size=img)
    o = OD.OracleDistillationV3(init["student_backbone"], scfg, teacher_state, tcfg, init["proj_global"], init["proj_local"], qsz, b, total,
                                weight_decay=float(oargs.weight_decay))
    assert (o.n_decay, o.n_no_decay) == tuple(len(g["params"]) for g in opt.param_groups), (o.n_decay, o.n_no_decay)
    steps = []
    for step in range(3):
        x = torch.randn(b, 3, img, img, generator=torch.Generator().manual_seed(2000 + step))
        torch.manual_seed(300 + step)
        lam = torch.empty(1).uniform_(0.0, 1.0).item()      # the draws of DistillationV3._mixup_data, in its order
        index = torch.randperm(b)
        rescales = None
        if s_kind == "dinov3":                              # then one RoPE rescale draw per student block (training mode)
            rescales = [torch.empty(1).uniform_(-math.log(2.0), math.log(2.0)).exp() for _ in range(2)]
        torch.manual_seed(300 + step)
        lr_now = opt.param_groups[0]["lr"]
        res = m.training_step_impl({"views": [x], "filename": []}, 0)
        res.loss.backward()
        params = [p for g in opt.param_groups for p in g["params"]]
        gnorm = torch.nn.utils.clip_grad_norm_(params, 1.0)       # configure_gradient_clipping: norm 1.0
        opt.step(); opt.zero_grad(set_to_none=True); sched.step()
        m.trainer.global_step += 1
        logs = {"loss": float(res.loss.detach()), "global_loss": res.log_dict["train_loss/global_loss"],
                "local_loss": res.log_dict["train_loss/local_loss"], "grad_norm": float(gnorm), "lr": lr_now}
        assert abs(o.opt.param_groups[0]["lr"] - lr_now) <= 1e-12 + 1e-6 * lr_now, (o.opt.param_groups[0]["lr"], lr_now)
        ol = o.train_step(x, lam, index, rescales)
        for k in ("loss", "global_loss", "local_loss", "grad_norm"):
            assert abs(ol[k] - logs[k]) <= 2e-5 * max(1.0, abs(logs[k])), (step, k, ol[k], logs[k])
        steps.append({"x_seed": 2000 + step, "lam": lam, "index": index.clone(), "logs": logs, "rescales": rescales})
        print(name, step, {k: round(v, 6) for k, v in logs.items()})
    final = {"student_backbone": {k: v.detach().clone() for k, v in s_model.state_dict().items()},
             "proj_global": {k: v.detach().clone() for k, v in m.student_projection_head_global.state_dict().items()},
             "proj_local": {k: v.detach().clone() for k, v in m.student_projection_head_local.state_dict().items()},
             "queue": m.teacher_queue.detach().clone()}
    osd = o.resnet.state_dict() if o.resnet is not None else o.sb    # parameters AND BatchNorm running statistics for the conv student
    for k, v in final["student_backbone"].items():
        assert (osd[k].detach().float() - v.float()).abs().max().item() <= 2e-6 + 2e-5 * v.float().abs().max().item(), k
    assert (o.queue - final["queue"]).abs().max().item() < 1e-6
    extra = {}
    if s_kind == "resnet":
        extra["reference_bf16"] = _reference_bf16_resnet_run(init, teacher_state, steps, final, b, img, total, qsz)
    torch.save({"b": b, "img": img, "total_steps": total, "queue_size": qsz, "weight_decay": float(oargs.weight_decay), "student_cfg": scfg,
                "teacher_cfg": tcfg, "teacher_state": teacher_state, "init": init, "steps": steps, "final": final, **extra},
               os.path.join(OUT, name + ".pt"))
    print("wrote", name, os.path.getsize(os.path.join(OUT, name + ".pt")) // 1024, "KiB")


def _reference_bf16_resnet_run(init, teacher_state, steps, final, b, img, total, qsz):
    """The reference's own DistillationV3 + ResNetModelWrapper once more from the same state and draws, but with training_step_impl
    under torch.autocast("cpu", bfloat16) (= precision "bf16-mixed"): how far the reference's own mixed-precision path is from its
    fp32 path on this fixture -- losses, grad-norm, and the fraction of parameter updates within 0.15 * lr * steps of the fp32 run
    (the measure tests/test_gpu_distill.py applies to the HIP step).  BatchNorm networks at random init amplify bf16 rounding:
    this is the yardstick for the tolerances of the convolutional student."""
    from lightly_train._methods.distillationv3.distillationv3 import DistillationV3, DistillationV3AdamWArgs, DistillationV3Args
    from lightly_train._models.dinov3.dinov3_src.models import vision_transformer as v3
    from lightly_train._models.dinov3.dinov3_vit import DINOv3ViTModelWrapper
    from lightly_train._models.embedding_model import EmbeddingModel
    from lightly_train._models.torchvision.resnet import ResNetModelWrapper
    from oracle import resnet_oracle as OR

    t = v3.DinoVisionTransformer(img_size=img, patch_size=16, embed_dim=64, depth=2, num_heads=1, ffn_ratio=4.0, qkv_bias=True,
                                 layerscale_init=0.5, norm_layer="layernormbf16", ffn_layer="mlp", n_storage_tokens=4, mask_k_bias=True,
                                 pos_embed_rope_base=100.0, pos_embed_rope_dtype="fp32", pos_embed_rope_rescale_coords=2)
    t.load_state_dict(teacher_state)
    s_model = OR.ResNet((1, 1, 1, 1), width=8)
    s_model.load_state_dict(init["student_backbone"])
    sw = ResNetModelWrapper(s_model)
    margs = DistillationV3Args(queue_size=qsz, teacher=DINOv3ViTModelWrapper(t))
    oargs = DistillationV3AdamWArgs()
    oargs.resolve_auto(wrapped_model=sw)
    m = DistillationV3(method_args=margs, optimizer_args=oargs, embedding_model=EmbeddingModel(wrapped_model=sw), global_batch_size=b,
                       num_input_channels=3)
    m.student_projection_head_global.load_state_dict(init["proj_global"])
    m.student_projection_head_local.load_state_dict(init["proj_local"])
    m.trainer = H.MockTrainer(total)
    [opt], [sched] = m.configure_optimizers()
    sched = sched["scheduler"]
    logs = []
    for step, rec in enumerate(steps):
        x = torch.randn(b, 3, img, img, generator=torch.Generator().manual_seed(rec["x_seed"]))
        torch.manual_seed(300 + step)
        with torch.autocast("cpu", dtype=torch.bfloat16):
            res = m.training_step_impl({"views": [x], "filename": []}, 0)
        res.loss.backward()
        gnorm = torch.nn.utils.clip_grad_norm_([p for g_ in opt.param_groups for p in g_["params"]], 1.0)
        opt.step(); opt.zero_grad(set_to_none=True); sched.step()
        m.trainer.global_step += 1
        logs.append({"loss": float(res.loss.detach()), "global_loss": float(res.log_dict["train_loss/global_loss"]),
                     "local_loss": float(res.log_dict["train_loss/local_loss"]), "grad_norm": float(gnorm)})
    lr = steps[-1]["logs"]["lr"]
    agree = tot = 0
    for k, v in final["student_backbone"].items():
        if k.startswith("fc.") or k.endswith(("running_mean", "running_var", "num_batches_tracked")):
            continue
        ours = s_model.state_dict()[k]
        if (v - init["student_backbone"][k]).abs().max().item() == 0:
            continue
        agree += int(((ours - v).abs() <= 0.15 * lr * len(steps)).sum()); tot += v.numel()
    out = {"logs": logs, "update_agreement": agree / tot,
           "max_rel_dev": {k: max(abs(a[k] - r["logs"][k]) / max(abs(r["logs"][k]), 1e-12) for a, r in zip(logs, steps)) for k in logs[0]}}
    print("reference bf16-autocast vs its fp32 run:", out["update_agreement"], out["max_rel_dev"])
    return out


def make_dinov3_vit() -> None:
    """(d) DINOv3 ViT forward (the distillation teacher, eval mode): the dinov3_vitl16 recipe (hub/backbones.py:467-512: RoPE
    base 100 in fp32, 4 storage tokens, LayerScale, eps 1e-5, K-masked qkv bias) at D=64, head_dim 64, 64^2 and 64x96 inputs."""
    H.install()
    from lightly_train._models.dinov3.dinov3_src.models import vision_transformer as v3
    from oracle import dinov3_oracle as O3

    torch.manual_seed(77)
    m = v3.DinoVisionTransformer(img_size=64, patch_size=16, embed_dim=64, depth=2, num_heads=1, ffn_ratio=4.0, qkv_bias=True,
                                 layerscale_init=0.5, norm_layer="layernormbf16", ffn_layer="mlp", n_storage_tokens=4, mask_k_bias=True,
                                 pos_embed_rope_base=100.0, pos_embed_rope_dtype="fp32", pos_embed_rope_rescale_coords=2)
    m.init_weights()
    for n_, prm in m.named_parameters():   # biases / LayerNorm affine are initialised to constants: randomise for a real test
        if n_.endswith(".bias") or "norm" in n_:
            prm.data.add_(0.1 * torch.randn_like(prm))
    m.eval()
    state = {k: v.detach().clone() for k, v in m.state_dict().items()}
    cfg = dict(patch_size=16, num_heads=1, depth=2, rope_base=100.0, ln_eps=1e-5, embed_dim=64, n_storage_tokens=4, img_size=64)
    cases = []
    for seed, (hh, ww) in ((5, (64, 64)), (6, (64, 96))):
        g = torch.Generator().manual_seed(seed)
        x = torch.randn(3, 3, hh, ww, generator=g)
        with torch.no_grad():
            ref = m.forward_features(x)
            mine = O3.dinov3_vit_forward(state, x, cfg)
        for k in ("x_norm_clstoken", "x_storage_tokens", "x_norm_patchtokens"):
            assert (ref[k] - mine[k]).abs().max().item() <= 2e-5 * max(1.0, ref[k].abs().max().item()), (k, (ref[k] - mine[k]).abs().max())
        cases.append({"seed": seed, "shape": (3, 3, hh, ww), "out": {k: ref[k].clone() for k in ("x_norm_clstoken", "x_storage_tokens", "x_norm_patchtokens")}})
    torch.save({"cfg": cfg, "state": state, "cases": cases}, os.path.join(OUT, "dinov3_vit_fwd.pt"))
    print("wrote dinov3_vit_fwd.pt", os.path.getsize(os.path.join(OUT, "dinov3_vit_fwd.pt")) // 1024, "KiB")


def make_reg4_swiglu() -> None:
    # (c) the shape of the reference's default pretrained family (dinov2_vit_src/configs: patch 14, 4 register tokens,
    # antialiased pos-embed interpolation with offset 0) + the SwiGLU FFN of the giant models, at D=64
    make_step_fixture("step_d64_reg4_swiglu14", "DinoVisionTransformer",
                      dict(embed_dim=64, depth=2, num_heads=1, mlp_ratio=4.0, num_register_tokens=4, ffn_layer="swiglu",
                           interpolate_antialias=True, interpolate_offset=0.0),
                      dict(output_dim=512, hidden_dim=128, dino_bottleneck_dim=64),
                      dict(patch_size=14, num_heads=1, depth=2, mlp_ratio=4.0, ffn_layer="swiglu", num_register_tokens=4,
                           interpolate_antialias=True, interpolate_offset=0.0),
                      b=8, g_size=56, l_size=28, n_local=2, n_steps=2, total_steps=50, keep_params_every_step=False)


if __name__ == "__main__":
    main()
