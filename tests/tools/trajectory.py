"""Loss trajectory of the HIP step against the fp32 CPU oracle over N optimizer steps on identical synthetic batches
(test infrastructure; north-star item "loss trajectory matching the reference over 100 synthetic steps").

Both sides start from the reference-generated initial state of tests/golden/step_d64_softmax.pt, see the same views
(seeded per step) and the same iBOT masks (the HIP step samples them with the reference's generator; the oracle is fed
the sampled masks).  Prints per-term deviations; `--koleo 0` removes the ill-conditioned KoLeo term."""
import argparse
import json
import os
import random
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def run(fixture: str, steps: int, koleo: float, lr_scale: float = 1.0, quiet: bool = False):
    import test_gpu_step as T

    fx = torch.load(os.path.join(ROOT, "tests", "golden", fixture + ".pt"), weights_only=False)
    fx = dict(fx, total_steps=max(fx["total_steps"], steps + 1))
    m = T.build(fx, koleo_loss_weight=koleo)
    o = T.oracle_for(fx, koleo_loss_weight=koleo)
    keys = ("loss", "dino_global_loss", "dino_local_loss", "ibot_loss", "koleo_loss")
    worst = {k: 0.0 for k in keys}
    rows = []
    for s in range(steps):
        views = T.synth_views(5000 + s, fx["b"], fx["g_size"], fx["l_size"], fx["n_local"])
        random.seed(900 + s)
        res = m.training_step_impl({"views": views}, s)
        masks = m._last_masks
        m.optimizer_step(); m.on_train_batch_end()
        ol = o.train_step(views, masks)
        ours = {k.split("/")[-1]: float(v) for k, v in res.log_dict.items()}
        ours["loss"] = float(res.loss)
        d = {k: abs(ours[k] - ol[k]) / max(1.0, abs(ol[k])) for k in keys}
        for k in keys:
            worst[k] = max(worst[k], d[k])
        rows.append((s, ours["loss"], ol["loss"], d))
        if not quiet and (s < 5 or s % 10 == 9):
            print(f"step {s:3d}  loss ours {ours['loss']:.5f}  oracle {ol['loss']:.5f}   rel dev " +
                  "  ".join(f"{k.split('_loss')[0]} {d[k]:.2e}" for k in keys))
    return worst, rows


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--fixture", default="step_d64_softmax")
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--koleo", type=float, default=0.1)
    a = ap.parse_args()
    worst, _ = run(a.fixture, a.steps, a.koleo)
    print(json.dumps({"fixture": a.fixture, "steps": a.steps, "koleo_weight": a.koleo, "max_rel_dev": worst}))
