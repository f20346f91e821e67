"""Diagnostic: ViTEngine forward/backward vs CPU autograd of the oracle ViT with a dense random upstream gradient."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import lightly_train_amd
from lightly_train_amd.vit import ViTConfig, ViTEngine, Workspace, init_vit_state, vit_param_shapes
from lightly_train_amd.params import FlatParams
from oracle import dinov2_oracle as O

def run(D, depth, heads, mlp_ratio, img, B, masked, ls=1e-5, seed=0):
    cfg = ViTConfig(embed_dim=D, depth=depth, num_heads=heads, mlp_ratio=mlp_ratio, patch_size=16, img_size=64, init_values=ls)
    g = torch.Generator().manual_seed(seed)
    sd = init_vit_state(cfg, g)
    for k in sd:  # make everything non-trivial
        if "bias" in k or k == "mask_token":
            sd[k] = torch.randn(sd[k].shape, generator=g) * 0.02
        if "gamma" in k:
            sd[k] = torch.rand(sd[k].shape, generator=g) * ls * 2
    fp = FlatParams([(n, sd[n]) for n, _ in vit_param_shapes(cfg)], "cuda", True)
    eng = ViTEngine(cfg, fp, "")
    ws = Workspace(torch.device("cuda"))
    x = torch.randn(B, 3, img, img, generator=g)
    n_p = (img // 16) ** 2
    masks = (torch.rand(B, n_p, generator=g) < 0.3) if masked else None
    ctx = eng.forward(ws, "s", x.cuda(), masks.to(torch.uint8).cuda() if masked else None, save=True)
    dxn = torch.randn(B, n_p + 1, D, generator=g)
    eng.backward(ws, ctx, dxn.cuda().contiguous())
    eng.finish_layerscale_grads()
    torch.cuda.synchronize()
    p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    out = O.vit_forward(p, x, dict(patch_size=16, num_heads=heads, depth=depth), masks=masks)
    xn_ref = torch.cat([out["cls"].unsqueeze(1), out["patch"]], 1)
    print(f"D={D} depth={depth} heads={heads} img={img} B={B} masked={masked} ls={ls}: fwd rel err",
          ((ctx['xn'].cpu() - xn_ref).abs().max() / xn_ref.abs().max()).item())
    (xn_ref * dxn).sum().backward()
    rows = []
    for n in fp.names:
        ref = p[n].grad
        if ref is None:
            continue
        rows.append(((fp.g[n].cpu() - ref).abs().max().item() / (ref.abs().max().item() + 1e-30), n, ref.abs().max().item()))
    rows.sort(reverse=True)
    for e, n, s in rows[:12]:
        print(f"   {e:10.3e} {n:40s} ref_max {s:.3e}")

run(64, 2, 1, 4.0, 64, 4, True, ls=1.0)
run(64, 2, 1, 4.0, 64, 4, True, ls=1e-5)
run(64, 2, 1, 4.0, 32, 4, False, ls=1.0)
run(8, 3, 2, 1.0, 64, 4, True, ls=1.0)
