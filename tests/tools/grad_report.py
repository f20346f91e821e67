"""Diagnostic: per-tensor gradient comparison HIP step vs the CPU oracle (fp32 autograd) on a golden fixture."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tools.parity_report import build_from_fixture, synth_views
from oracle import dinov2_oracle as O

name = sys.argv[1] if len(sys.argv) > 1 else "step_vittest_softmax"
fx = torch.load(os.path.join(ROOT, "tests", "golden", name + ".pt"), weights_only=False)
kw = float(os.environ.get("KOLEO_W", "0.1"))
m = build_from_fixture(fx)
m.method_args.koleo_loss_weight = kw
mk = fx["method_kwargs"]
o = O.OracleDINOv2(fx["init"]["student_backbone"], fx["init"]["student_head"], fx["cfg"],
                   args=dict(output_dim=mk["output_dim"], hidden_dim=mk["hidden_dim"], bottleneck_dim=mk["dino_bottleneck_dim"],
                             center_method=mk.get("center_method", "softmax"), koleo_loss_weight=kw),
                   global_batch_size=fx["b"], total_steps=fx["total_steps"], teacher_head=fx["init"]["teacher_head"])
rec = fx["steps"][0]
views = synth_views(rec["view_seed"], fx["b"], fx["g_size"], fx["l_size"], fx["n_local"])
res = m.training_step_impl({"views": views}, 0, masks=rec["masks"])
torch.cuda.synchronize()
loss, logs = o.forward_loss(views, rec["masks"])
loss.backward()
print("loss ours", float(res.loss), "oracle", float(loss))
rows = []
for n in m.student.names:
    ref = (o.sb[n[9:]] if n.startswith("backbone.") else o.sh[n[5:]]).grad
    ours = m.student.g[n].cpu()
    scale = ref.abs().max().item() + 1e-30
    rows.append((((ours - ref).abs().max().item()) / scale, n, scale, ours.abs().max().item()))
rows.sort(reverse=True)
for e, n, s, om in rows:
    print(f"{e:10.3e}  {n:55s} ref_max {s:10.3e} ours_max {om:10.3e}")
