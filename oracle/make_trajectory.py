"""TEST INFRASTRUCTURE ONLY.  100-optimizer-step loss trajectories written by the REFERENCE's own DINOv2 class
(imported from /root/reference through oracle/ref_harness.py, Lightning hook order of ReferenceRunner) -- the north-star
item "loss trajectory matching the reference to 1e-3 over 100 synthetic steps".

For each KoLeo weight (0.1 = reference default, 0.0) three runs from the SAME initial state, views and iBOT masks:
  * "fp32"      the reference on CPU in fp32                                     -> the trajectory the HIP step is held to
  * "bf16"      the reference under torch.autocast("cpu", bfloat16) around training_step_impl, i.e. what Lightning's
                precision="bf16-mixed" does (forward under autocast, fp32 master weights / optimizer) -> how far the
                reference's OWN mixed-precision path drifts from its fp32 path on this trajectory
  * "fp32_perturbed"  fp32 with the initial student weights perturbed by 1 ulp-scale noise (relative 1e-7) -> the
                trajectory's sensitivity to rounding at the fp32 level (chaos floor of the KoLeo nearest-neighbour term)

Run in the build container:  python -m oracle.make_trajectory      (writes tests/golden/trajectory_d64.pt, ~70 KiB;
the initial state is the one of tests/golden/step_d64_softmax.pt -- asserted here)
The HIP step reproduces views (seed 5000 + s) and masks (random.seed(900 + s) before each step: the mask sampler is
bit-identical to the reference's, tests/test_host_logic.py) on the GPU box; nothing else of the reference travels.
"""
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
KEYS = ("loss", "dino_global_loss", "dino_local_loss", "ibot_loss", "koleo_loss", "grad_norm")
CFG = dict(arch="DinoVisionTransformer", model_kwargs=dict(embed_dim=64, depth=2, num_heads=1, mlp_ratio=4.0),
           method_kwargs=dict(output_dim=512, hidden_dim=128, dino_bottleneck_dim=64),
           cfg=dict(patch_size=16, num_heads=1, depth=2), b=8, g_size=96, l_size=48, n_local=2)
# `--config mid` (round 4): a mid-size model -- D = 192, 3 heads of 64, 4 blocks, K = 4096 prototypes, 2 x 112^2 + 4 x 48^2 crops, batch 8 --
# whose initial state is NOT the reference constructor's draw but this package's seeded initialisers (init_vit_state / init_head_state,
# seed below) loaded into the reference module, so the fixture carries a seed instead of 15 MB of weights and the GPU box rebuilds the
# same tensors.  LayerScale starts at 1.0: at the default 1e-5 the cls tokens of a batch agree to 1e-5 and the KoLeo term is chaotic from
# step 0; at 1.0 the "KoLeo on" column says something about the kernels.
MID = dict(arch="DinoVisionTransformer", model_kwargs=dict(embed_dim=192, depth=4, num_heads=3, mlp_ratio=4.0, init_values=1.0),
           method_kwargs=dict(output_dim=4096, hidden_dim=512, dino_bottleneck_dim=128),
           cfg=dict(patch_size=16, num_heads=3, depth=4), b=8, g_size=112, l_size=48, n_local=4, init_seed=2024, init_values=1.0)


# `--config vits` (round 5): ViT-S width -- D = 384, 6 heads of 64, 6 blocks, K = 16 384 prototypes, head 2048 / 256 (the reference's default
# head widths), 2 x 112^2 + 4 x 48^2 crops, batch 8: 800 global / 320 local token rows, i.e. the 256-row four-phase GEMM, the slab split-K
# weight gradients and the register-resident wide-row softmax / cross-entropy kernels all take part.  Seeded state like `mid`.
VITS = dict(arch="DinoVisionTransformer", model_kwargs=dict(embed_dim=384, depth=6, num_heads=6, mlp_ratio=4.0, init_values=1.0),
            method_kwargs=dict(output_dim=16384, hidden_dim=2048, dino_bottleneck_dim=256),
            cfg=dict(patch_size=16, num_heads=6, depth=6), b=8, g_size=112, l_size=48, n_local=4, init_seed=2025, init_values=1.0)
# `--config vitb` (round 6): the HEADLINE model -- ViT-B/16: D = 768, 12 heads of 64, 12 blocks, K = 65 536 prototypes, head 2048 / 256, 2 x 224^2 +
# 8 x 96^2 crops (the benchmark's crop geometry with the local crops at the multiple of the patch size the reference class accepts), batch 8:
# 3152 global / 2304 local token rows per pass.  Seeded state like `mid` / `vits`, LayerScale 1.0 for the same reason.  Six runs of 100 steps are
# hours of CPU in the build container, so every (KoLeo weight, mode) run is its own restartable piece (`--piece koleo mode`, cached under
# tests/golden/.trajectory_vitb_parts/) and `--config vitb` without `--piece` assembles the fixture from the pieces that exist.
VITB = dict(arch="DinoVisionTransformer", model_kwargs=dict(embed_dim=768, depth=12, num_heads=12, mlp_ratio=4.0, init_values=1.0),
            method_kwargs=dict(output_dim=65536, hidden_dim=2048, dino_bottleneck_dim=256),
            cfg=dict(patch_size=16, num_heads=12, depth=12), b=8, g_size=224, l_size=96, n_local=8, init_seed=2026, init_values=1.0)
CONFIGS = {"d64": CFG, "mid": MID, "vits": VITS, "vitb": VITB}


def seeded_init(cfg):
    """The initial state of the `mid` configuration: (student backbone, student head, teacher head) from one seeded generator."""
    import lightly_train_amd  # noqa: F401
    from lightly_train_amd.dinov2 import init_head_state
    from lightly_train_amd.vit import ViTConfig, init_vit_state

    mk, mkw = cfg["method_kwargs"], cfg["model_kwargs"]
    g = torch.Generator().manual_seed(cfg["init_seed"])
    vc = ViTConfig(embed_dim=mkw["embed_dim"], depth=mkw["depth"], num_heads=mkw["num_heads"], mlp_ratio=mkw["mlp_ratio"], patch_size=16,
                   img_size=cfg["g_size"], init_values=cfg["init_values"])
    bsd = init_vit_state(vc, g)
    shs = init_head_state(mkw["embed_dim"], mk["hidden_dim"], mk["dino_bottleneck_dim"], mk["output_dim"], g)
    ths = init_head_state(mkw["embed_dim"], mk["hidden_dim"], mk["dino_bottleneck_dim"], mk["output_dim"], g)
    return vc, bsd, shs, ths


def load_seeded_init(m, cfg) -> None:
    """Overwrite the reference module's freshly constructed weights with the seeded initial state (teacher backbone = student backbone)."""
    _, bsd, shs, ths = seeded_init(cfg)
    sd = m.state_dict()
    new = {}
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


def run(koleo: float, steps: int, mode: str, init_state=None, CFG=CFG):
    """One trajectory of the reference class.  Returns (per-step logs, initial split state)."""
    mk = dict(CFG["method_kwargs"], koleo_loss_weight=koleo)
    m = H.build_reference_method(arch=CFG["arch"], patch_size=16, img_size=CFG["g_size"], model_kwargs=CFG["model_kwargs"],
                                 method_kwargs=mk, global_batch_size=CFG["b"], total_steps=steps + 1, seed=1234)
    if "init_seed" in CFG:
        load_seeded_init(m, CFG)
    if mode == "fp32_perturbed":
        g = torch.Generator().manual_seed(99)
        with torch.no_grad():
            for p in m.student_embedding_model.parameters():
                p.mul_(1.0 + 1e-7 * torch.randn(p.shape, generator=g))
    r = H.ReferenceRunner(m)
    init = r.split_state()
    if mode == "bf16":
        inner = m.training_step_impl

        def autocast_step(batch, batch_idx):
            with torch.autocast("cpu", dtype=torch.bfloat16):
                return inner(batch, batch_idx)

        m.training_step_impl = autocast_step
    rows = []
    for s in range(steps):
        views = synth_views(5000 + s, CFG["b"], CFG["g_size"], CFG["l_size"], CFG["n_local"])
        random.seed(900 + s)
        logs = r.train_step(views)
        rows.append({k: float(logs[k]) for k in KEYS})
        if s < 3 or s % 20 == 19:
            print(mode, koleo, s, {k: round(v, 5) for k, v in rows[-1].items()}, flush=True)
    return rows, init


def main() -> None:
    steps = 100
    name = sys.argv[sys.argv.index("--config") + 1] if "--config" in sys.argv else "d64"
    cfg = CONFIGS[name]
    mid = "init_seed" in cfg     # seeded initial state (mid, vits)
    out = {"cfg": cfg, "steps": steps, "view_seed0": 5000, "mask_seed0": 900, "runs": {}}
    parts = os.path.join(OUT, f".trajectory_{name}_parts")
    if "--piece" in sys.argv:      # one restartable run: python -m oracle.make_trajectory --config vitb --piece 0.1 bf16
        i = sys.argv.index("--piece")
        koleo, mode = float(sys.argv[i + 1]), sys.argv[i + 2]
        os.makedirs(parts, exist_ok=True)
        rows, init = run(koleo, steps, mode, CFG=cfg)
        if mode == "fp32" and mid:
            _, bsd, shs, ths = seeded_init(cfg)
            for part, want in (("student_backbone", bsd), ("student_head", shs), ("teacher_head", ths)):
                for k, v in init[part].items():
                    assert torch.equal(v, want[k].reshape(v.shape)), (part, k)
        torch.save(rows, os.path.join(parts, f"{koleo}_{mode}.pt"))
        print("wrote piece", koleo, mode)
        return
    if os.path.isdir(parts):       # assemble the fixture from the pieces that exist (fp32 is required per KoLeo weight)
        for koleo in (0.1, 0.0):
            for mode in ("fp32", "bf16", "fp32_perturbed"):
                f = os.path.join(parts, f"{koleo}_{mode}.pt")
                if os.path.exists(f):
                    out["runs"][(koleo, mode)] = torch.load(f, weights_only=False)
        summary = {}
        for (koleo, mode), alt in out["runs"].items():
            if mode != "fp32":
                ref = out["runs"][(koleo, "fp32")]
                summary[(koleo, mode)] = {k: max(abs(a[k] - b[k]) / max(1.0, abs(b[k])) for a, b in zip(alt, ref)) for k in KEYS}
                print("max rel dev vs fp32", koleo, mode, {k: f"{v:.2e}" for k, v in summary[(koleo, mode)].items()})
        out["summary"] = summary
        path = os.path.join(OUT, f"trajectory_{name}.pt")
        torch.save(out, path)
        print("wrote", path, os.path.getsize(path) // 1024, "KiB", sorted(out["runs"]))
        return
    for koleo in (0.1, 0.0):
        for mode in ("fp32", "bf16", "fp32_perturbed"):
            rows, init = run(koleo, steps, mode, CFG=cfg)
            out["runs"][(koleo, mode)] = rows
            if mode == "fp32" and mid:   # the reference module really holds the seeded state the GPU box will rebuild
                _, bsd, shs, ths = seeded_init(cfg)
                for part, want in (("student_backbone", bsd), ("student_head", shs), ("teacher_head", ths)):
                    for k, v in init[part].items():
                        assert torch.equal(v, want[k].reshape(v.shape)), (part, k)
            if mode == "fp32" and not mid:   # same seed, same config => the initial state of tests/golden/step_d64_softmax.pt (not stored twice)
                fx = torch.load(os.path.join(OUT, "step_d64_softmax.pt"), weights_only=False)
                for part in ("student_backbone", "student_head", "teacher_head"):
                    for k, v in init[part].items():
                        assert torch.equal(v, fx["init"][part][k]), (part, k)
    # deviations of the reference's own alternative paths from its fp32 path
    summary = {}
    for koleo in (0.1, 0.0):
        ref = out["runs"][(koleo, "fp32")]
        for mode in ("bf16", "fp32_perturbed"):
            alt = out["runs"][(koleo, mode)]
            summary[(koleo, mode)] = {k: max(abs(a[k] - b[k]) / max(1.0, abs(b[k])) for a, b in zip(alt, ref)) for k in KEYS}
            print("max rel dev vs fp32", koleo, mode, {k: f"{v:.2e}" for k, v in summary[(koleo, mode)].items()})
    out["summary"] = summary
    path = os.path.join(OUT, f"trajectory_{name}.pt")
    torch.save(out, path)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
