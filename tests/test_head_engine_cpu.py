"""CPU: the ORCHESTRATION of the projection-head engine (lightly_train_amd/dinov2.py::HeadEngine) in exact arithmetic -- plain-torch
stand-ins for the HIP ops (tests/tools/ops_emu.py), fp32 buffers -- against torch autograd of the reference's own DINOv2ProjectionHead
(LT/_methods/dinov2/dinov2_head.py), imported from /root/reference when it is present (the build container).  Covers what the bf16 GPU
comparisons can only see loosely: the BatchNorm variant's per-call statistics over row segments, the order in which the calls move the
running estimates, eval-mode teacher heads, GELU' / BatchNorm backward wiring, bias column sums, the zero-padded weight-gradient
contractions and the weight-norm gradients."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))

import lightly_train_amd  # noqa: E402,F401
import ops_emu  # noqa: E402
from lightly_train_amd import ops  # noqa: E402
from lightly_train_amd.dinov2 import DINOv2Args, HeadEngine, head_param_shapes, init_head_state  # noqa: E402
from lightly_train_amd.params import FlatParams  # noqa: E402
from lightly_train_amd.vit import Workspace  # noqa: E402
from oracle import ref_harness as H  # noqa: E402

pytestmark = pytest.mark.skipif(not H.reference_available(), reason="reference tree not present")


class F32Workspace(Workspace):
    def get(self, name, shape, dtype, **kw):
        return super().get(name, shape, torch.float32 if dtype == torch.bfloat16 else dtype, **kw)


def build(use_bn: bool, D=24, hid=40, bott=16, K=72, seed=0):
    H.install()
    from lightly_train._methods.dinov2.dinov2_head import DINOv2ProjectionHead
    torch.manual_seed(seed)
    ref = DINOv2ProjectionHead(in_dim=D, out_dim=K, use_bn=use_bn, hidden_dim=hid, bottleneck_dim=bott)
    with torch.no_grad():
        for n, p in ref.named_parameters():          # biases / BatchNorm affine / weight-norm g start at constants: move them
            p.add_(0.1 * torch.randn_like(p))
    sd = {k: v.detach().clone() for k, v in ref.state_dict().items()}
    args = DINOv2Args(output_dim=K, hidden_dim=hid, dino_bottleneck_dim=bott, batch_norm=use_bn)
    named = [("head." + n, sd[n]) for n, _ in head_param_shapes(D, hid, bott, K, use_bn)]
    fp = FlatParams(named, "cpu", True)
    fp.bf16 = fp.data.clone()                        # "bf16 shadow" = the fp32 weights themselves
    fp.b = {n: fp.bf16[fp.offsets[n]:fp.offsets[n] + fp.p[n].numel()].view(fp.shapes[n]) for n in fp.names}
    eng = HeadEngine(fp, "head.", D, args)
    eng.wn = eng.wn.float()
    eng.load_buffers(sd)
    eng.refresh_weightnorm()
    return ref, eng, fp, sd


@pytest.mark.parametrize("use_bn", [False, True])
def test_head_engine_forward_backward_equals_the_reference_head(use_bn):
    D, K = 24, 72
    segs = [(0, 6), (6, 9), (15, 70)]                 # rows of three separate calls of the reference head; 85 rows: 43 short of 128
    R = sum(n for _, n in segs)
    cap = 128
    with ops_emu.emulate(ops):
        ref, eng, fp, sd = build(use_bn)
        g = torch.Generator().manual_seed(3)
        x = torch.zeros(cap, D); x[:R] = torch.randn(R, D, generator=g)
        d = torch.zeros(cap, K); d[:R] = torch.randn(R, K, generator=g) * 0.1
        ws = F32Workspace(torch.device("cpu"))
        order = [segs[0], segs[2], segs[1]]           # the reference's call order (global cls, masked patches, local cls)
        c = eng.forward(ws, "h", x, R, cap, save=True, segs=order)
        logits = c["logits"][:R].clone()
        fp.grad.zero_()
        dx = eng.backward(ws, c, d.clone())[:R].clone()
        eng.finish_weightnorm_grad()
    ref.train()
    xr = x[:R].clone().requires_grad_(True)
    outs = {}
    for r0, n in order:
        outs[r0] = ref(xr[r0:r0 + n])
    out = torch.cat([outs[r0] for r0, _ in segs])
    (out * d[:R]).sum().backward()
    assert torch.allclose(logits, out.detach(), atol=2e-5), (logits - out.detach()).abs().max()
    assert torch.allclose(dx, xr.grad, atol=2e-5, rtol=1e-4), (dx - xr.grad).abs().max()
    for n, p in ref.named_parameters():
        mine = fp.g["head." + n]
        assert torch.allclose(mine, p.grad, atol=3e-5, rtol=2e-4), (n, (mine - p.grad).abs().max().item())
    if use_bn:
        bufs = eng.buffer_state()
        for k, v in ref.state_dict().items():
            if k.endswith(("running_mean", "running_var")):
                assert torch.allclose(bufs[k], v, atol=1e-6), k
            elif k.endswith("num_batches_tracked"):
                assert int(bufs[k]) == int(v) == 3, k


def test_batchnorm_head_in_eval_mode_applies_the_running_estimates():
    """The teacher heads as the reference constructs them (freeze_eval_module, dinov2.py:63-67,241)."""
    D, K, R, cap = 24, 72, 21, 64
    with ops_emu.emulate(ops):
        ref, eng, fp, sd = build(True, seed=5)
        with torch.no_grad():                         # running estimates away from (0, 1)
            for k, v in ref.state_dict().items():
                if k.endswith("running_mean"):
                    v.add_(0.3 * torch.randn_like(v))
                elif k.endswith("running_var"):
                    v.mul_(1.0 + 0.5 * torch.rand_like(v))
        eng.load_buffers(ref.state_dict())
        x = torch.zeros(cap, D); x[:R] = torch.randn(R, D, generator=torch.Generator().manual_seed(1))
        ws = F32Workspace(torch.device("cpu"))
        c = eng.forward(ws, "t", x, R, cap, save=False, segs=[(0, 8), (8, 13)], bn_training=False)
        logits = c["logits"][:R].clone()
    ref.eval()
    with torch.no_grad():
        out = torch.cat([ref(x[:8]), ref(x[8:R])])
    assert torch.allclose(logits, out, atol=2e-5)
    assert all(int(v) == 0 for k, v in eng.buffer_state().items() if k.endswith("num_batches_tracked"))
