"""Test infrastructure (not a product path): the fixture of the configuration `bench.py` times.

One step of the pinned fp32 restatement (oracle/dinov2_oracle.py, itself checked against the reference's own class on the smaller
fixtures: tests/test_oracle_pin.py) at the BENCHMARK's shapes -- ViT-B/16 (D = 768, 12 blocks, LayerScale 1e-5), K = 65 536 prototypes,
2 x 224^2 + 8 x 98^2 crops, softmax centering, drop-path 0 -- at the largest batch the build container's 62 GB hold for an autograd step
(default 32: 6336 global / 12 800 local token rows, so every token GEMM is on the 256-row four-phase kernel and every weight gradient on
the slab split-K kernel, with the 65 536-wide register-resident softmax / cross-entropy kernels).  LT/_methods/dinov2/dinov2.py:259-397.

Stored (tests/golden/bench_vitb_b<batch>.pt, ~2 MB, not the state): the seeds that rebuild weights and views bit for bit, the iBOT masks,
and from the step with KoLeo OFF (per-tensor gradients are well conditioned there, see tests/test_gpu_step.py): loss terms, total gradient
norm, per-tensor gradient norms of EVERY parameter, 14 named gradient tensors (small ones whole, matrices as a strided sample), the two
loss centers after the step's update; from a forward with the reference's default KoLeo weight: the four loss terms.

    python oracle/make_bench_fixture.py [--batch 32] [--threads 8]

`--reference --local-size 96 [--batch 16] [--arch vit_small]` (round 5; round 6: `--arch vit_small --batch 32` = BASELINE configs[1], ViT-S/16, at a
quarter of its batch -> tests/golden/bench_vits_ref_b32.pt): the SAME quantities written by the REFERENCE's own DINOv2 class
(oracle/ref_harness.py imports it from /root/reference; its wrapper refuses 98^2 crops at patch 16, SURVEY 8(d), so the local crops are the
upstream default 96^2 = 37 tokens) -> tests/golden/bench_vitb_ref_b<batch>.pt.  The seeded weights are loaded into the reference module
(asserted equal), the masks are the ones its own `create_collated_masks` call sampled, the gradients are autograd's on its parameters.
This pins ViT-B/16 with K = 65 536 on the reference itself rather than on the restatement.
"""
from __future__ import annotations

import argparse
import os
import random
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lightly_train_amd  # noqa: E402,F401
from lightly_train_amd.dinov2 import DINOv2Args, init_head_state  # noqa: E402
from lightly_train_amd.masking import MaskingGenerator, create_collated_masks  # noqa: E402
from lightly_train_amd.vit import ViTConfig, init_vit_state  # noqa: E402
from oracle import dinov2_oracle as O  # noqa: E402

SAMPLED = ["cls_token", "pos_embed", "mask_token", "patch_embed.proj.weight", "blocks.0.norm1.weight", "blocks.0.attn.qkv.weight", "blocks.0.attn.qkv.bias",
           "blocks.5.mlp.fc1.weight", "blocks.5.mlp.fc1.bias", "blocks.5.ls2.gamma", "blocks.11.attn.proj.weight", "blocks.11.mlp.fc2.weight", "norm.weight",
           "head.mlp.0.weight", "head.mlp.4.bias", "head.last_layer.parametrizations.weight.original1"]


def sample(t: torch.Tensor) -> torch.Tensor:
    """Small tensors whole; matrices as every 16th row x every 8th column (the test applies the same stride)."""
    if t.numel() <= 1 << 16:
        return t.clone()
    m = t.reshape(t.shape[0], -1) if t.dim() > 1 and t.shape[0] > 1 else t.reshape(-1, t.shape[-1])
    return m[::16, ::8].clone()


ARCHS = {"vit_base": (768, 12), "vit_small": (384, 6)}     # embed_dim, heads (12 blocks each): BASELINE configs[2] / configs[1]


def build_inputs(batch: int, seed: int, local: int = 98, arch: str = "vit_base"):
    """Weights and views from one generator, in this order (tests/test_gpu_step.py rebuilds them the same way)."""
    g = torch.Generator().manual_seed(seed)
    D, H = ARCHS[arch]
    vc = ViTConfig(embed_dim=D, depth=12, num_heads=H, mlp_ratio=4.0, patch_size=16, img_size=224, init_values=1e-5)
    bsd = init_vit_state(vc, g)
    shs, ths = init_head_state(D, 2048, 256, 65536, g), init_head_state(D, 2048, 256, 65536, g)
    views = [torch.randn(batch, 3, 224, 224, generator=g) for _ in range(2)] + [torch.randn(batch, 3, local, local, generator=g) for _ in range(8)]
    return vc, bsd, shs, ths, views


def reference_step(b: int, seed: int, mask_seed: int, local: int, koleo, backward: bool, autocast: bool = False, arch: str = "vit_base"):
    """One `training_step_impl` (+ backward) of the reference's own class from the seeded state.  Returns a dict in the layout of the
    restatement's record ("loss", "logs", and with `backward`: grad_norm / tensor_norms / grad_samples / centers / logit_samples / masks)."""
    from oracle import ref_harness as H

    H.install()
    import lightly_train._methods.dinov2.dinov2 as ref_dinov2

    vc, bsd, shs, ths, views = build_inputs(b, seed, local, arch)
    mk = dict(output_dim=65536)
    if koleo is not None:
        mk["koleo_loss_weight"] = koleo
    m = H.build_reference_method(arch=arch, patch_size=16, img_size=224, method_kwargs=mk, global_batch_size=b, total_steps=100, seed=1)
    sd, new = m.state_dict(), {}
    for k, v in sd.items():
        for role, head in (("student", shs), ("teacher", ths)):
            pre = f"{role}_embedding_model.wrapped_model._model."
            if k.startswith(pre):
                new[k] = bsd[k[len(pre):]].reshape(v.shape)
            for hn in ("dino_head", "ibot_head"):
                pre = f"{role}_head.{hn}."
                if k.startswith(pre):
                    new[k] = head[k[len(pre):]].reshape(v.shape)
        new.setdefault(k, v)
    m.load_state_dict(new, strict=True)
    assert torch.equal(m.state_dict()["student_embedding_model.wrapped_model._model.blocks.3.mlp.fc1.weight"], bsd["blocks.3.mlp.fc1.weight"])
    cap: dict = {}
    t_calls, s_calls = [], []

    def spy(mod, sink):
        orig = mod.forward

        def fwd(x):
            out = orig(x)
            sink.append(out.detach()[:4, ::64].float().clone())
            return out
        mod.forward = fwd

    spy(m.teacher_head.dino_head, t_calls)
    spy(m.student_head.dino_head, s_calls)
    orig_ccm = ref_dinov2.create_collated_masks

    def spy_ccm(**kw):
        out = orig_ccm(**kw)
        cap["masks"] = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in out.items()}
        return out

    ref_dinov2.create_collated_masks = spy_ccm
    random.seed(mask_seed)
    t0 = time.time()
    try:
        if autocast:
            with torch.autocast("cpu", dtype=torch.bfloat16):
                res = m.training_step_impl({"views": views, "filename": []}, 0)
        elif backward:
            res = m.training_step_impl({"views": views, "filename": []}, 0)
        else:
            with torch.no_grad():
                res = m.training_step_impl({"views": views, "filename": []}, 0)
    finally:
        ref_dinov2.create_collated_masks = orig_ccm
    rec = {"loss": float(res.loss.detach()), "logs": {k.split("/")[-1]: float(v) for k, v in res.log_dict.items()}, "masks": cap["masks"]}
    if not backward:
        return rec
    res.loss.backward()
    print(f"reference forward + backward ({'bf16 autocast' if autocast else 'fp32'}): {time.time() - t0:.1f} s", flush=True)
    norms, samples, sq = {}, {}, 0.0
    for role_pre, ours in (("student_embedding_model.wrapped_model._model.", "backbone."), ("student_head.dino_head.", "head.")):
        for n, p in m.named_parameters():
            if not n.startswith(role_pre) or p.grad is None:
                continue
            short = n[len(role_pre):]
            g2 = float((p.grad.double() ** 2).sum())
            sq += g2
            norms[ours + short] = g2 ** 0.5
            if (short if ours == "backbone." else ours + short) in SAMPLED:
                samples[ours + short] = sample(p.grad.float())
    rec.update(grad_norm=sq ** 0.5, tensor_norms=norms, grad_samples=samples)
    m.dino_loss.apply_center_update()
    m.ibot_loss.apply_center_update()
    rec["dino_center"] = m.dino_loss.center.detach().reshape(-1).clone()
    rec["ibot_center"] = m.ibot_loss.center.detach().reshape(-1).clone()
    # head calls in the reference's order: teacher (cls, masked patches), student (cls, masked patches, local cls)
    rec["logit_samples"] = {"t_cls_logits": t_calls[0], "t_patch_logits": t_calls[1], "s_cls_logits": s_calls[0], "s_patch_logits": s_calls[1],
                            "s_loc_logits": s_calls[2]}
    return rec


def main_reference(a) -> None:
    b = a.batch
    out = {"batch": b, "seed": a.seed, "mask_seed": a.mask_seed, "total_steps": 100, "local_size": a.local_size, "writer": "reference class", "arch": a.arch,
           "config": f"{a.arch}/16, K=65536, 2x224^2 + 8x{a.local_size}^2, softmax centering -- LT/_methods/dinov2/dinov2.py:259-397 run from /root/reference"}
    k0 = reference_step(b, a.seed, a.mask_seed, a.local_size, 0.0, backward=True, arch=a.arch)
    out["masks"] = k0.pop("masks")
    out["koleo0"] = k0
    yb = reference_step(b, a.seed, a.mask_seed, a.local_size, 0.0, backward=True, autocast=True, arch=a.arch)
    yard = {"loss": yb["loss"], "logs": yb["logs"], "grad_norm": yb["grad_norm"], "norm_rel_err": {}, "sample_err": {}}
    for n, v in yb["tensor_norms"].items():
        yard["norm_rel_err"][n] = abs(v - k0["tensor_norms"][n]) / max(k0["tensor_norms"][n], 1e-20)
    for n, v in yb["grad_samples"].items():
        ref = k0["grad_samples"][n]
        yard["sample_err"][n] = float((v.reshape(ref.shape) - ref).abs().max() / (ref.abs().max() + 1e-20))
    out["autocast_yardstick"] = yard
    d = reference_step(b, a.seed, a.mask_seed, a.local_size, None, backward=False, arch=a.arch)
    assert all(torch.equal(d["masks"][k], out["masks"][k]) for k in out["masks"] if torch.is_tensor(out["masks"][k]))
    out["default"] = {"loss": d["loss"], "logs": d["logs"]}
    path = os.path.join(ROOT, "tests", "golden", f"bench_{'vitb' if a.arch == 'vit_base' else 'vits'}_ref_b{b}.pt")
    torch.save(out, path)
    print(path, os.path.getsize(path) // 1024, "KiB", out["koleo0"]["logs"], out["default"]["logs"], "grad-norm", out["koleo0"]["grad_norm"],
          "autocast grad-norm", yard["grad_norm"])


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--seed", type=int, default=404)
    ap.add_argument("--mask-seed", type=int, default=17)
    ap.add_argument("--threads", type=int, default=os.cpu_count())
    ap.add_argument("--reference", action="store_true", help="write the fixture with the reference's own class (needs /root/reference; --local-size 96)")
    ap.add_argument("--local-size", type=int, default=98)
    ap.add_argument("--arch", default="vit_base", choices=sorted(ARCHS), help="--reference only: vit_small = BASELINE configs[1] (DINOv2 ViT-S/16) -> bench_vits_ref_b<batch>.pt")
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    if a.reference:
        if a.local_size % 16:
            raise SystemExit("the reference's wrapper cannot run local crops that are not a multiple of the patch size (SURVEY 8(d)): use --local-size 96")
        return main_reference(a)
    b = a.batch
    vc, bsd, shs, ths, views = build_inputs(b, a.seed)
    cfg = dict(patch_size=16, num_heads=12, depth=12)
    out = {"batch": b, "seed": a.seed, "mask_seed": a.mask_seed, "total_steps": 100, "config": "vit_base/16, K=65536, 2x224^2 + 8x98^2, softmax centering"}

    # ---- step with KoLeo off: losses, gradients, centers
    o = O.OracleDINOv2(bsd, shs, cfg, args=dict(koleo_loss_weight=0.0), global_batch_size=b, total_steps=100, teacher_head=ths)
    # iBOT masks from the product's host sampler (bit-identical to the reference's MaskingGenerator: tests/test_host_logic.py), seeded
    da = DINOv2Args()
    random.seed(a.mask_seed)
    gen = MaskingGenerator(input_size=(14, 14), max_num_patches=int(0.5 * 14 * 14))
    masks = create_collated_masks(da.mask_ratio_min, da.mask_ratio_max, int(2 * b * da.mask_probability), 2 * b, gen)
    t0 = time.time()
    cap: dict = {}
    loss, logs = o.forward_loss(views, masks, capture=cap)
    loss.backward()
    print(f"forward + backward: {time.time() - t0:.1f} s", flush=True)
    out["masks"] = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in masks.items()}
    out["koleo0"] = {"loss": float(loss.detach()), "logs": {k: float(v) for k, v in logs.items()}}
    norms, samples, sq = {}, {}, 0.0
    for name, p in [("backbone." + n, p) for n, p in o.sb.items()] + [("head." + n, p) for n, p in o.sh.items()]:
        gnorm2 = float((p.grad.double() ** 2).sum())
        sq += gnorm2
        norms[name] = gnorm2 ** 0.5
        short = name[9:] if name.startswith("backbone.") else name
        if short in SAMPLED:
            samples[name] = sample(p.grad)
    out["koleo0"].update(grad_norm=sq ** 0.5, tensor_norms=norms, grad_samples=samples)
    # the teacher logits' statistics the step leaves behind: the centers after the update that the NEXT step applies (dinov2_loss.py:139-160)
    o._apply_center_updates()
    out["koleo0"]["dino_center"] = o.dino_center.detach().reshape(-1).clone()
    out["koleo0"]["ibot_center"] = o.ibot_center.detach().reshape(-1).clone()
    out["koleo0"]["logit_samples"] = {k: cap[k].detach()[:4, ::64].clone() for k in ("t_cls_logits", "s_cls_logits", "s_loc_logits", "s_patch_logits", "t_patch_logits")}
    fp32_samples = samples
    fp32_norms = norms
    del o, loss, cap

    # ---- the yardstick column: the SAME restatement under torch.autocast("cpu", bfloat16) (what precision="bf16-mixed" does to the reference:
    # bf16 matmul operands, fp32 master weights) against its own fp32 run -- per-tensor gradient errors of a bf16 pipeline that is not ours
    ob = O.OracleDINOv2(bsd, shs, cfg, args=dict(koleo_loss_weight=0.0), global_batch_size=b, total_steps=100, teacher_head=ths)
    t0 = time.time()
    with torch.autocast("cpu", dtype=torch.bfloat16):
        loss_b, logs_b = ob.forward_loss(views, masks)
    loss_b.backward()
    print(f"bf16-autocast forward + backward: {time.time() - t0:.1f} s", flush=True)
    yard = {"loss": float(loss_b.detach()), "logs": {k: float(v) for k, v in logs_b.items()}, "norm_rel_err": {}, "sample_err": {}}
    sqb = 0.0
    for name, p in [("backbone." + n, p) for n, p in ob.sb.items()] + [("head." + n, p) for n, p in ob.sh.items()]:
        nb = float(p.grad.double().norm())
        sqb += nb * nb
        yard["norm_rel_err"][name] = abs(nb - fp32_norms[name]) / max(fp32_norms[name], 1e-20)
        if name in fp32_samples:
            ref = fp32_samples[name]
            yard["sample_err"][name] = float((sample(p.grad.float()).reshape(ref.shape) - ref).abs().max() / (ref.abs().max() + 1e-20))
    yard["grad_norm"] = sqb ** 0.5
    out["autocast_yardstick"] = yard
    del ob, loss_b

    # ---- forward with the reference's defaults (KoLeo 0.1), same weights / views / masks
    o2 = O.OracleDINOv2(bsd, shs, cfg, args={}, global_batch_size=b, total_steps=100, teacher_head=ths)
    with torch.no_grad():
        loss2, logs2 = o2.forward_loss(views, masks)
    out["default"] = {"loss": float(loss2), "logs": {k: float(v) for k, v in logs2.items()}}
    path = os.path.join(ROOT, "tests", "golden", f"bench_vitb_b{b}.pt")
    torch.save(out, path)
    print(path, os.path.getsize(path) // 1024, "KiB", out["koleo0"]["logs"], out["default"]["logs"], "grad-norm", out["koleo0"]["grad_norm"])


if __name__ == "__main__":
    main()
