"""Debug aid (GPU box): run the HIP ResNet engine and its CPU bf16 emulation side by side and print, layer by layer, the first
place where their intermediates diverge (forward order, then backward order)."""
import contextlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
import lightly_train_amd  # noqa: E402,F401
import ops_emu  # noqa: E402
from lightly_train_amd import ops  # noqa: E402
from lightly_train_amd.params import FlatParams  # noqa: E402
from lightly_train_amd.resnet import ARCHS, ResNetConfig, ResNetEngine, flat_named, init_resnet_state  # noqa: E402
from lightly_train_amd.vit import Workspace  # noqa: E402


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-20)).item()


def main(arch="_resnet_test", B=8, S=64):
    cfg = ResNetConfig(**ARCHS[arch])
    g = torch.Generator().manual_seed(5)
    sd = init_resnet_state(cfg, g)
    for k in sd:
        if (".bn" in k or k.startswith("bn") or "downsample.1" in k) and k.endswith(("weight", "bias")):
            sd[k] = sd[k] + 0.2 * torch.randn(sd[k].shape, generator=g)
    x = torch.randn(B, 3, S, S, generator=g)
    runs = {}
    d = None
    for dev in ("cuda", "cpu"):
        with (ops_emu.emulate(ops) if dev == "cpu" else contextlib.nullcontext()):
            fp = FlatParams(flat_named(cfg, sd), dev, True)
            eng = ResNetEngine(cfg, fp, "", buffers=sd)
            ws = Workspace(torch.device(dev))
            ctx = eng.forward(ws, "r", x.to(dev), save=True, train=True)
            n = B * ctx["h"] * ctx["w"]
            if d is None:
                d = torch.randn(n, cfg.feature_dim, generator=g) * 0.1
            dfeat = torch.zeros_like(ctx["feat"])
            dfeat[:n] = d.to(torch.bfloat16).to(dev)
            fp.grad.zero_()
            eng.backward(ws, ctx, dfeat)
            if dev == "cuda":
                torch.cuda.synchronize()
            runs[dev] = (ctx, {k: v.float().cpu().clone() for k, v in ws.bufs.items() if v.dtype != torch.uint8}, {k: fp.g[k].float().cpu().clone() for k in fp.names})
    (ch, bh, gh), (ce, be, ge) = runs["cuda"], runs["cpu"]
    s_h, s_e = ch["stem"], ce["stem"]
    r1, r2 = s_h["r1"], s_h["r2"]
    print(f"== {arch} B={B} S={S}")
    print("stem cols", rel(s_h["cols"][:r1], s_e["cols"][:r1]), " c", rel(s_h["c"][:r1], s_e["c"][:r1]), " a", rel(s_h["a"], s_e["a"]),
          " mean", rel(s_h["mean"], s_e["mean"]), " rstd", rel(s_h["rstd"], s_e["rstd"]), " pool", rel(bh["r.stem.pool"][:r2], be["r.stem.pool"][:r2]))
    for rh, re_ in zip(ch["blocks"], ce["blocks"]):
        rin, rout = rh["rin"], rh["rout"]
        line = [rh["prefix"], "c1 %.1e" % rel(rh["c1"][:rin], re_["c1"][:rin]), "a1 %.1e" % rel(rh["a1"], re_["a1"]),
                "cols2 %.1e" % rel(rh["cols2"][:rout], re_["cols2"][:rout]), "c2 %.1e" % rel(rh["c2"][:rout], re_["c2"][:rout]),
                "a2 %.1e" % rel(rh["a2"][:rout], re_["a2"][:rout]), "c3 %.1e" % rel(rh["c3"][:rout], re_["c3"][:rout])]
        if rh["down"]:
            line += ["xs %.1e" % rel(rh["xs"][:rout], re_["xs"][:rout]), "cd %.1e" % rel(rh["cd"][:rout], re_["cd"][:rout])]
        line += ["out %.1e" % rel(rh["out"][:rout], re_["out"][:rout]), "m3 %.1e" % rel(rh["m3"], re_["m3"]), "s3 %.1e" % rel(rh["s3"], re_["s3"])]
        print(" ".join(line))
    print("-- backward: per-block persistent buffers")
    for k in sorted(bh):
        if any(t in k for t in (".dz3", ".dc3", ".dc2", ".dc1", ".dcd", ".dx", "stem.dc", "stem.da", "stem.dz")) and k in be and bh[k].shape == be[k].shape:
            print(f"{k:32s} {rel(bh[k], be[k]):.1e}")
    worst = sorted(((rel(gh[k], ge[k]), k) for k in gh), reverse=True)
    print("-- parameter gradients: worst", [(f"{r:.2e}", k) for r, k in worst[:6]], "median %.2e" % worst[len(worst) // 2][0])


if __name__ == "__main__":
    main("_resnet_test", 8, 64)
    main("resnet50", 4, 64)
