"""CPU: the ORCHESTRATION of the convolutional student engine (lightly_train_amd/resnet.py) -- layouts ([Cout][kh][kw][Cin] weights,
NHWC rows), buffer shapes, the order and wiring of the explicit backward through stem, max-pool, bottlenecks, downsample and
identity paths -- against torch autograd of the restated torchvision ResNet (oracle/resnet_oracle.py) in fp32, with plain-torch
stand-ins for the HIP ops (tests/tools/ops_emu.py).  The kernels behind those ops are checked on the MI355X (tests/test_gpu_ops.py,
tests/test_gpu_distill.py::test_resnet_engine_matches_bf16_emulation)."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))

import lightly_train_amd  # noqa: E402,F401
import ops_emu  # noqa: E402
from lightly_train_amd import ops  # noqa: E402
from lightly_train_amd.params import FlatParams  # noqa: E402
from lightly_train_amd.resnet import (ARCHS, ResNetConfig, ResNetEngine, flat_named, from_flat_layout, init_resnet_state,  # noqa: E402
                                      state_dict_order)
from lightly_train_amd.vit import Workspace  # noqa: E402
from oracle import resnet_oracle as OR  # noqa: E402


def rel(a, b):
    return ((a.float() - b.float()).abs().max() / (b.float().abs().max() + 1e-20)).item()


def fp32_engine(cfg, sd):
    fp = FlatParams(flat_named(cfg, sd), "cpu", True)
    eng = ResNetEngine(cfg, fp, "", buffers=sd)
    eng.act_dtype = torch.float32                       # exact arithmetic: isolates the orchestration from bf16 rounding
    eng.w_stem = eng.w_stem.float()
    eng.refresh_padded_weights()
    fp.bf16 = fp.data.clone()
    fp.b = {n: fp.bf16[fp.offsets[n]:fp.offsets[n] + fp.p[n].numel()].view(fp.shapes[n]) for n in fp.names}
    return fp, eng


@pytest.mark.parametrize("layers,width,B,S,seed", [((2, 1, 1, 1), 8, 4, 64, 11), ((1, 2, 1, 2), 8, 2, 96, 3)])
def test_resnet_engine_forward_backward_equals_torch_autograd(layers, width, B, S, seed):
    cfg = ResNetConfig(layers=layers, width=width)
    g = torch.Generator().manual_seed(seed)
    sd = init_resnet_state(cfg, g)
    for k in sd:
        if (".bn" in k or k.startswith("bn") or "downsample.1" in k) and k.endswith(("weight", "bias")):
            sd[k] = sd[k] + 0.2 * torch.randn(sd[k].shape, generator=g)
    x = torch.randn(B, 3, S, S, generator=g)
    m = OR.ResNet(layers, width=width)
    m.load_state_dict(sd)
    m.train()
    fm = OR.features(m, x)
    with ops_emu.emulate(ops):
        fp, eng = fp32_engine(cfg, sd)
        ws = Workspace(torch.device("cpu"))
        ctx = eng.forward(ws, "r", x, save=True, train=True)
        h, w, C = ctx["h"], ctx["w"], cfg.feature_dim
        n = B * h * w
        assert fm.shape == (B, C, h, w)
        assert rel(ctx["feat"][:n].view(B, h, w, C).permute(0, 3, 1, 2), fm) < 1e-4
        d = torch.randn(n, C, generator=g) * 0.1
        fm.backward(d.view(B, h, w, C).permute(0, 3, 1, 2))
        dfeat = torch.zeros_like(ctx["feat"])
        dfeat[:n] = d
        fp.grad.zero_()
        eng.backward(ws, ctx, dfeat)
        # a ReLU input within rounding of zero may land on either side in the two implementations: allow one such channel
        errs = {nm: rel(from_flat_layout(nm, fp.g[nm]), p.grad) for nm, p in m.named_parameters() if not nm.startswith("fc.")}
        bad = {k: v for k, v in errs.items() if not v < 1e-3}
        assert len(bad) <= 2, bad
        # running statistics and the torchvision-layout export
        out = eng.state_dict(extra={"fc.weight": sd["fc.weight"], "fc.bias": sd["fc.bias"]})
        assert list(out) == state_dict_order(cfg) == list(m.state_dict())
        for k, v in m.state_dict().items():
            if k.endswith(("running_mean", "running_var")):
                assert rel(out[k], v) < 1e-4, k
            elif k.endswith("num_batches_tracked"):
                assert int(out[k]) == int(v) == 1
            else:
                assert torch.equal(out[k], sd[k]), k
        # eval mode uses the running estimates
        m.eval()
        with torch.no_grad():
            fe = OR.features(m, x)
        eng.load_state_dict(m.state_dict())
        ce = eng.forward(ws, "e", x, save=False, train=False)
        assert rel(ce["feat"][:n].view(B, h, w, C).permute(0, 3, 1, 2), fe) < 1e-4


def test_resnet_shapes_tables():
    cfg = ResNetConfig(**ARCHS["resnet50"])
    assert cfg.feature_dim == 2048 and len(state_dict_order(cfg)) == 320
