"""TEST INFRASTRUCTURE ONLY.  tests/golden/dinov31_d64.pt: three optimizer steps of the REFERENCE's own `DINOv31` class
(LT/_methods/dinov31/dinov31.py, imported from /root/reference through oracle/ref_harness.py, Lightning hook order incl. the EMA of the
PaKA head) on a D = 64 / depth-2 ViT with `paka_start_step = 1` (step 0 is plain DINOv2, steps 1-2 carry the PaKA term), 2 + 2 DINO views,
2 clean globals and 4 PaKA locals per image, synthetic crop geometries with flips and one image whose local does not overlap its parent.

`lightly.loss.PatchKernelAlignmentLoss` / `roi_resample_to_grid` are not in the reference tree: the class runs on the restatements of
oracle/dinov31_oracle.py (PARITY UNPINNED for those two definitions).  Pinned by this fixture: everything dinov31.py does around them.

    python -m oracle.make_dinov31_fixture
"""
from __future__ import annotations

import os
import random
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref_harness as H  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
B, G_SIZE, L_SIZE, N_LOCAL, K_PAKA = 8, 96, 48, 2, 4
KEYS = ("dino_global_loss", "dino_local_loss", "ibot_loss", "koleo_loss", "paka_loss")


def synth_batch(seed: int):
    """Views in the DINOv31 layout and their geometries [B, 8] = (x0, y0, x1, y1, image_w, image_h, hflip, vflip)."""
    g = torch.Generator().manual_seed(seed)
    views = [torch.randn(B, 3, G_SIZE, G_SIZE, generator=g) for _ in range(2)] + [torch.randn(B, 3, L_SIZE, L_SIZE, generator=g) for _ in range(N_LOCAL)]
    views += [torch.randn(B, 3, G_SIZE, G_SIZE, generator=g) for _ in range(2)] + [torch.randn(B, 3, L_SIZE, L_SIZE, generator=g) for _ in range(K_PAKA)]
    rng = random.Random(seed)
    W, Hh = 500.0, 375.0
    gbox = []
    for v in range(2):
        rows = []
        for b in range(B):
            w, h = rng.uniform(180, 360), rng.uniform(150, 300)
            x0, y0 = rng.uniform(0, W - w), rng.uniform(0, Hh - h)
            rows.append([round(x0), round(y0), round(x0 + w), round(y0 + h), W, Hh, float(rng.random() < 0.5), 0.0])
        gbox.append(torch.tensor(rows, dtype=torch.float32))
    geoms = [gbox[0], gbox[1]]
    for _ in range(N_LOCAL):          # the DINO locals' geometry is recorded but not read by the method
        geoms.append(torch.tensor([[10.0, 10.0, 110.0, 110.0, W, Hh, 0.0, 0.0]] * B))
    geoms += [gbox[0].clone(), gbox[1].clone()]     # the clean globals re-use the globals' geometry verbatim (constrained_crop.py:64-107)
    for k in range(K_PAKA):
        par = gbox[k % 2]
        rows = []
        for b in range(B):
            px0, py0, px1, py1 = (float(par[b, i]) for i in range(4))
            pw, ph = px1 - px0, py1 - py0
            area = rng.uniform(0.05, 0.40) * pw * ph
            asp = rng.uniform(0.75, 4.0 / 3.0)
            w, h = min(max(1.0, round((area * asp) ** 0.5)), pw), min(max(1.0, round((area / asp) ** 0.5)), ph)
            x0, y0 = px0 + rng.randint(0, int(pw - w)), py0 + rng.randint(0, int(ph - h))
            if k == 3 and b == 0:     # one pair without a shared region: excluded from the loss's mean
                x0, y0 = (px1 + 5.0 if px1 + 5.0 + w < W else max(0.0, px0 - w - 5.0)), py0
            rows.append([x0, y0, x0 + w, y0 + h, W, Hh, float(rng.random() < 0.5), float(rng.random() < 0.25)])
        geoms.append(torch.tensor(rows, dtype=torch.float32))
    return views, geoms


def sample(t: torch.Tensor) -> torch.Tensor:
    """Small tensors whole, matrices as every 16th row x every 8th column."""
    return t.clone() if t.numel() <= 4096 else t.reshape(t.shape[0], -1)[::16, ::8].clone()


def main() -> None:
    H.install()
    from lightly_train._methods.dinov2.dinov2 import DINOv2AdamWViTArgs
    from lightly_train._methods.dinov2 import utils as ref_utils
    import lightly_train._methods.dinov2.dinov2 as ref_dinov2
    from lightly_train._methods.dinov31.dinov31 import DINOv31, DINOv31Args
    from lightly_train._models.dinov2_vit.dinov2_vit import DINOv2ViTModelWrapper
    from lightly_train._models.dinov2_vit.dinov2_vit_src.models import vision_transformer as vits
    from lightly_train._models.embedding_model import EmbeddingModel

    torch.manual_seed(4321)
    random.seed(4321)
    model = vits.DinoVisionTransformer(img_size=G_SIZE, patch_size=16, embed_dim=64, depth=2, num_heads=1, mlp_ratio=4.0, init_values=1.0,
                                       drop_path_rate=0.0, ffn_layer="mlp", block_chunks=0, interpolate_offset=0.1)
    wrapped = DINOv2ViTModelWrapper(model)
    mk = dict(output_dim=512, hidden_dim=128, dino_bottleneck_dim=64, koleo_loss_weight=0.0, paka_num_local=K_PAKA, paka_start_step=1, paka_weight=0.7)
    margs = DINOv31Args(**mk)
    oargs = DINOv2AdamWViTArgs()
    margs.resolve_auto(scaling_info=None, optimizer_args=oargs, wrapped_model=wrapped)
    total_steps = 20
    m = DINOv31(method_args=margs, optimizer_args=oargs, embedding_model=EmbeddingModel(wrapped_model=wrapped), global_batch_size=B, num_input_channels=3)
    m.trainer = H.MockTrainer(total_steps)
    # the PaKA heads are 64 -> 2048 -> 2048 -> 256 whatever the backbone (hard-coded widths, dinov31.py:131-140): 4.8 M parameters each.  They
    # start from this package's seeded initialiser instead of the constructor's draw, so that the fixture carries a seed, not 19 MB
    import lightly_train_amd  # noqa: F401
    from lightly_train_amd.dinov31 import init_paka_head_state

    paka_seed = 99
    ph = init_paka_head_state(64, torch.Generator().manual_seed(paka_seed))
    m.student_paka_head.load_state_dict(ph)
    m.teacher_paka_head.load_state_dict(ph)
    r = H.ReferenceRunner(m)
    init = r.split_state()
    fixture = {"cfg": dict(patch_size=16, num_heads=1, depth=2, init_values=1.0), "method_kwargs": mk, "b": B, "g_size": G_SIZE, "l_size": L_SIZE,
               "n_local": N_LOCAL, "k_paka": K_PAKA, "total_steps": total_steps,
               "init": {"student_backbone": init["student_backbone"], "student_head": init["student_head"], "teacher_head": init["teacher_head"]},
               "paka_seed": paka_seed, "steps": []}
    cap: dict = {}
    orig_ccm = ref_utils.create_collated_masks

    def spy_ccm(**kw):
        out = orig_ccm(**kw)
        cap["masks"] = {k: v.clone() for k, v in out.items()}
        return out

    ref_dinov2.create_collated_masks = spy_ccm
    for step in range(3):
        views, geoms = synth_batch(7000 + step)
        batch = {"views": views, "filename": [], "geometries": geoms}
        random.seed(300 + step)
        res = m.training_step_impl(batch, step)
        res.loss.backward()
        m.on_before_optimizer_step(r.optim)
        params = [p for g in r.optim.param_groups for p in g["params"]]
        paka_grad = {k: sample(v.grad.detach()) for k, v in m.student_paka_head.named_parameters() if v.grad is not None}      # (before clipping)
        paka_gnorm = {k: float(v.grad.norm()) for k, v in m.student_paka_head.named_parameters() if v.grad is not None}
        gnorm = torch.nn.utils.clip_grad_norm_(params, m.method_args.gradient_clip_val)
        r.optim.step()
        r.optim.zero_grad(set_to_none=True)
        r.sched.step()
        m.trainer.global_step += 1
        try:
            m.on_train_batch_end(None, batch, step)
        except Exception:
            pass
        logs = {k.split("/")[-1]: float(v) for k, v in res.log_dict.items()}
        logs["loss"], logs["grad_norm"] = float(res.loss.detach()), float(gnorm)
        rec = {"seed": 7000 + step, "geometries": geoms, "masks": cap["masks"], "logs": logs, "paka_grad": paka_grad, "paka_grad_norm": paka_gnorm}
        if step == 2:     # the final state: backbone / projection heads / centers whole, the PaKA heads as a strided sample + norms
            sd = m.state_dict()
            rec["state"] = {k: v.detach().clone() for k, v in sd.items() if "_paka_head." not in k}
            rec["paka_state"] = {k: sample(v.detach()) for k, v in sd.items() if "_paka_head." in k}
            rec["paka_state_norm"] = {k: float(v.norm()) for k, v in sd.items() if "_paka_head." in k}
        fixture["steps"].append(rec)
        print(step, {k: round(v, 6) for k, v in logs.items()})
    ref_dinov2.create_collated_masks = orig_ccm
    assert "paka_loss" not in fixture["steps"][0]["logs"] and fixture["steps"][1]["logs"]["paka_loss"] > 0
    path = os.path.join(OUT, "dinov31_d64.pt")
    torch.save(fixture, path)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
