"""Diagnostic (test infrastructure): run the HIP DINOv2 step on the golden fixtures (generated from the
reference's own code, oracle/make_golden.py) and print per-quantity deviations."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import lightly_train_amd  # noqa: E402
from lightly_train_amd.dinov2 import DINOv2, DINOv2Args  # noqa: E402
from lightly_train_amd.vit import ViTConfig  # noqa: E402


def synth_views(seed, b, g_size, l_size, n_local):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(b, 3, g_size, g_size, generator=g) for _ in range(2)] + [
        torch.randn(b, 3, l_size, l_size, generator=g) for _ in range(n_local)]


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-20)).item()


def build_from_fixture(fx, device="cuda"):
    cfgd, mk = fx["cfg"], fx["method_kwargs"]
    sb = fx["init"]["student_backbone"]
    D = sb["cls_token"].shape[-1]
    hid = sb["blocks.0.mlp.fc1.weight"].shape[0]
    vc = ViTConfig(embed_dim=D, depth=cfgd["depth"], num_heads=cfgd["num_heads"], mlp_ratio=hid / D, patch_size=cfgd["patch_size"],
                   img_size=fx["g_size"])
    args = DINOv2Args(output_dim=mk.get("output_dim", 65536), hidden_dim=mk.get("hidden_dim", 2048),
                      dino_bottleneck_dim=mk.get("dino_bottleneck_dim", 256), center_method=mk.get("center_method", "softmax"),
                      ibot_separate_head=mk.get("ibot_separate_head", False))
    m = DINOv2(vc, args, global_batch_size=fx["b"], total_steps=fx["total_steps"], device=device, backbone_state=sb,
               student_head_state=fx["init"]["student_head"], teacher_head_state=fx["init"]["teacher_head"],
               student_ibot_head_state=fx["init"].get("student_ibot_head"), teacher_ibot_head_state=fx["init"].get("teacher_ibot_head"))
    return m


def main():
    for name in ("step_vittest_softmax", "step_vittest_sinkhorn", "step_d64_softmax"):
        fx = torch.load(os.path.join(ROOT, "tests", "golden", name + ".pt"), weights_only=False)
        m = build_from_fixture(fx)
        print("==", name)
        for si, rec in enumerate(fx["steps"]):
            views = synth_views(rec["view_seed"], fx["b"], fx["g_size"], fx["l_size"], fx["n_local"])
            assert abs(float(sum(v.double().sum() for v in views)) - rec["view_checksum"]) < 1e-6
            res = m.training_step_impl({"views": views}, 0, masks=rec["masks"])
            torch.cuda.synchronize()
            L = m._last
            B, M = L["B"], L["M"]
            print(f" step {si}: teacher cls logits rel {rel(L['t_cls_logits'], rec['teacher_cls_logits']):.3e}  patch {rel(L['t_patch_logits'], rec['teacher_patch_logits']):.3e}")
            print(f"          student cls logits rel {rel(L['s_cls_logits'], rec['student_cls_logits']):.3e}  patch {rel(L['s_patch_logits'], rec['student_patch_logits']):.3e}  local {rel(L['s_local_logits'], rec['student_local_logits']):.3e}")
            logs = {k.split('/')[-1]: float(v) for k, v in res.log_dict.items()}
            logs["loss"] = float(res.loss)
            print("          losses ours/ref:", {k: (round(logs[k], 5), round(rec["logs"][k], 5)) for k in logs})
            m.optimizer_step()
            gn = float(m.last_grad_norm.sqrt())
            print(f"          grad_norm ours {gn:.4f} ref {rec['logs']['grad_norm']:.4f}")
            m.on_train_batch_end()
            if "state" in rec:
                st = rec["state"]
                for role, fp in (("student", m.student), ("teacher", m.teacher)):
                    worst = ("", 0.0)
                    for n in fp.names:
                        ref = st[f"{role}_backbone"][n[9:]] if n.startswith("backbone.") else st[f"{role}_head"][n[5:]]
                        init = fx["init"]["student_backbone"][n[9:]] if n.startswith("backbone.") else fx["init"][f"{role}_head"][n[5:]]
                        upd = (ref - init).abs().max().item() + 1e-12
                        e = (fp.p[n].cpu() - ref).abs().max().item() / upd
                        if e > worst[1]:
                            worst = (n, e)
                    print(f"          {role} params: worst |ours-ref|/|ref-init| = {worst[1]:.3e} at {worst[0]}")
                print(f"          centers: dino {rel(m.dino_center, rec['dino_center']):.3e} ibot {rel(m.ibot_center, rec['ibot_center']):.3e} (applied lazily: compare after next step)")


if __name__ == "__main__":
    main()
