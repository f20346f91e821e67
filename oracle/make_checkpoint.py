"""TEST INFRASTRUCTURE ONLY.  tests/golden/ckpt_d64.pt: a training checkpoint written around the REFERENCE's own DINOv2 module
(imported from /root/reference through oracle/ref_harness.py) -- `method.state_dict()`, `torch.optim.AdamW.state_dict()` of its
fused parameter groups, scheduler position and global step after TWO optimizer steps -- plus what the reference computes in the
step that follows a resume (losses, grad-norm, a checksum of every updated student / teacher tensor).  The HIP method must
load it (DINOv2.load_checkpoint_dict), export it back bit-identically and reproduce step three.

Run in the build container:  python -m oracle.make_checkpoint"""
from __future__ import annotations

import os
import random
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref_harness as H  # noqa: E402
from oracle.make_golden import synth_views  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def main() -> None:
    H.install()
    import lightly_train._methods.dinov2.dinov2 as ref_dinov2
    from lightly_train._methods.dinov2 import utils as ref_utils

    b, g_size, l_size, n_local, total = 8, 96, 48, 2, 50
    # KoLeo off: its in-branch gradients are ill-conditioned at initialisation (tests/golden/trajectory_d64.pt: a 1e-7 perturbation of the
    # fp32 reference moves its own trajectory by 2e-3) and this fixture is about the resume mechanics, not about that term
    mk = dict(output_dim=512, hidden_dim=128, dino_bottleneck_dim=64, student_freeze_backbone_steps=1, koleo_loss_weight=0.0)
    m = H.build_reference_method(arch="DinoVisionTransformer", patch_size=16, img_size=g_size,
                                 model_kwargs=dict(embed_dim=64, depth=2, num_heads=1, mlp_ratio=4.0), method_kwargs=mk,
                                 global_batch_size=b, total_steps=total, seed=4321)
    r = H.ReferenceRunner(m)
    cap: dict = {}
    orig = ref_utils.create_collated_masks

    def spy(**kw):
        out = orig(**kw)
        cap["masks"] = {k: v.clone() for k, v in out.items()}
        return out

    ref_dinov2.create_collated_masks = spy
    pre_logs = []
    for s in range(2):
        random.seed(310 + s)
        pre_logs.append(r.train_step(synth_views(3000 + s, b, g_size, l_size, n_local)))
    # student_freeze_backbone_steps=1: the backbone must not have moved in step 0 (lr = 0 there), the head must have
    ckpt = {"state_dict": {k: v.detach() for k, v in m.state_dict().items()},   # aliases of the shared head stay aliases (one storage)
            "optimizer_states": [r.optim.state_dict()],
            "lr_schedulers": [{k: v for k, v in r.sched.state_dict().items() if k != "lr_lambdas"}],
            "global_step": m.trainer.global_step, "epoch": 0}
    ckpt = torch.load(_roundtrip(ckpt), weights_only=False)   # detach the saved tensors from the live optimizer state
    # ---- a true resume: a FRESH reference module (other seed) restored from the checkpoint alone, as Lightning does on `ckpt_path=`
    # (module state, optimizer state, scheduler position, global step).  What is not in a checkpoint is gone on both sides -- notably the
    # DINO / iBOT center update that was still pending (computed at step 2, applied lazily at the start of step 3, dinov2_loss.py:135-160).
    m2 = H.build_reference_method(arch="DinoVisionTransformer", patch_size=16, img_size=g_size,
                                  model_kwargs=dict(embed_dim=64, depth=2, num_heads=1, mlp_ratio=4.0), method_kwargs=mk,
                                  global_batch_size=b, total_steps=total, seed=999)
    r2 = H.ReferenceRunner(m2)
    import copy

    loaded = copy.deepcopy(ckpt)     # torch's Optimizer.load_state_dict may alias the tensors it is given ("step" is then bumped in place)
    m2.load_state_dict(loaded["state_dict"])
    r2.optim.load_state_dict(loaded["optimizer_states"][0])
    r2.sched.load_state_dict(dict(ckpt["lr_schedulers"][0], lr_lambdas=[{} for _ in r2.optim.param_groups]))
    m2.trainer.global_step = ckpt["global_step"]
    random.seed(312)
    logs = r2.train_step(synth_views(3002, b, g_size, l_size, n_local))
    random.seed(312)
    live = r.train_step(synth_views(3002, b, g_size, l_size, n_local))    # the uninterrupted run, for the record (differs: pending centers)
    m, r = m2, r2
    keep_t = ("teacher_embedding_model.wrapped_model._model.blocks.1.mlp.fc1.weight", "teacher_head.dino_head.mlp.0.weight",
              "teacher_head.dino_head.last_layer.parametrizations.weight.original1")
    after = {k: v.detach() for k, v in m.state_dict().items() if ".ibot_head." not in k and (k.startswith("student_") or k in keep_t or "center" in k)}
    fixture = {"cfg": dict(patch_size=16, num_heads=1, depth=2, embed_dim=64), "method_kwargs": mk, "b": b, "g_size": g_size, "l_size": l_size,
               "n_local": n_local, "total_steps": total, "pre_logs": pre_logs, "checkpoint": ckpt,
               "step3": {"view_seed": 3002, "masks": cap["masks"], "logs": logs, "logs_uninterrupted_run": live, "state_after": after,
                         "optimizer_after": r.optim.state_dict()["state"][0]}}
    path = os.path.join(OUT, "ckpt_d64.pt")
    torch.save(fixture, path)
    print("pre", pre_logs)
    print("step3", logs)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def _roundtrip(obj):
    import io

    buf = io.BytesIO()
    torch.save(obj, buf)
    buf.seek(0)
    return buf


if __name__ == "__main__":
    main()
