"""-m gpu: the plain-torch statements of the kernel contracts that the exact-arithmetic CPU tests run on (tests/tools/ops_emu.py) against
the kernels themselves, for the fused ops whose semantics carry the orchestration (scales, row subsets, fused by-products): same random
inputs through the HIP wrapper on the MI355X and through the stand-in on the CPU, outputs and by-products compared at the bf16 level.
This closes the loop of DESIGN 3: kernel == contract (here and in tests/test_gpu_ops.py), contracts compose to the reference (CPU)."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))


def _both(call, tensors, outs):
    """Run `call(ops, T)` with T = the tensors on the GPU (real wrappers) and on the CPU (stand-ins); returns {name: (gpu, cpu)} for `outs`."""
    import lightly_train_amd  # noqa: F401
    import ops_emu
    from lightly_train_amd import ops

    dev = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in tensors.items()}
    call(ops, dev)
    torch.cuda.synchronize()
    cpu = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in tensors.items()}
    with ops_emu.emulate(ops):
        call(ops, cpu)
    return {k: (dev[k].float().cpu(), cpu[k].float()) for k in outs}


def _close(pair, rtol=2e-2, atol=2e-2):
    a, b = pair
    scale = b.abs().max().item() + 1e-12
    assert (a - b).abs().max().item() <= atol * scale + 1e-6, ((a - b).abs().max().item(), scale)


def test_gemm_residual_epilogue_with_row_and_branch_scale():
    g = torch.Generator().manual_seed(0)
    M, N, K = 300, 64, 128
    T = dict(a=torch.randn(M, K, generator=g).bfloat16(), b=(torch.randn(N, K, generator=g) * 0.1).bfloat16(), out=torch.zeros(M, N), out2=torch.zeros(M, N).bfloat16(),
             bias=torch.randn(N, generator=g), gamma=torch.randn(N, generator=g), resid=torch.randn(M, N, generator=g), rs=torch.rand(M, generator=g))
    r = _both(lambda ops, t: ops.gemm(t["a"], t["b"], t["out"], M=M, N=N, K=K, epilogue=ops.EPI_RESID, bias=t["bias"], gamma=t["gamma"], resid=t["resid"],
                                     out2=t["out2"], rowscale=t["rs"], branch_scale=1.75), T, ["out", "out2"])
    _close(r["out"]); _close(r["out2"])


def test_gemm_gelu_and_gelugrad_epilogues():
    g = torch.Generator().manual_seed(1)
    M, N, K = 260, 96, 64
    T = dict(a=torch.randn(M, K, generator=g).bfloat16(), b=(torch.randn(N, K, generator=g) * 0.2).bfloat16(), out=torch.zeros(M, N).bfloat16(),
             pre=torch.zeros(M, N).bfloat16(), bias=torch.randn(N, generator=g), aux=(torch.randn(M, N, generator=g) * 2).bfloat16(), dout=torch.zeros(M, N).bfloat16())
    r = _both(lambda ops, t: (ops.gemm(t["a"], t["b"], t["out"], M=M, N=N, K=K, epilogue=ops.EPI_BF16_GELU, bias=t["bias"], out2=t["pre"]),
                              ops.gemm(t["a"], t["b"], t["dout"], M=M, N=N, K=K, epilogue=ops.EPI_BF16_GELUGRAD, aux=t["aux"])), T, ["out", "pre", "dout"])
    for k in r:
        _close(r[k])


def test_fused_layernorm_backward_and_layerscale():
    g = torch.Generator().manual_seed(2)
    R, D = 200, 64
    x = torch.randn(R, D, generator=g)
    T = dict(x=x, w=torch.randn(D, generator=g), mean=x.mean(1), rstd=(x.var(1, unbiased=False) + 1e-6).rsqrt(), dy=torch.randn(R, D, generator=g).bfloat16(),
             dres=torch.randn(R, D, generator=g), dx=torch.zeros(R, D), dw=torch.zeros(D), db=torch.zeros(D), dnext=torch.zeros(R, D).bfloat16(),
             gn=torch.randn(D, generator=g), rsn=torch.rand(R, generator=g), dbn=torch.zeros(D),
             y=torch.randn(R, D, generator=g).bfloat16(), gam=torch.randn(D, generator=g), dy2=torch.zeros(R, D).bfloat16(), dgam=torch.zeros(D), dbias=torch.zeros(D))
    r = _both(lambda ops, t: (ops.layernorm_bwd(t["x"], t["w"], t["mean"], t["rstd"], t["dy"], t["dres"], t["dx"], t["dw"], t["db"], R, D, dnext=t["dnext"],
                                                gamma_next=t["gn"], rowscale_next=t["rsn"], scale_next=1.5, dbias_next=t["dbn"]),
                              ops.layerscale_bwd(t["dres"], t["y"], t["gam"], t["dy2"], t["dgam"], R, D, dbias=t["dbias"], rowscale=t["rsn"], scale=0.75)),
              T, ["dx", "dw", "db", "dnext", "dbn", "dy2", "dgam", "dbias"])
    for k in r:
        _close(r[k])


def test_token_assembly_and_its_backward():
    g = torch.Generator().manual_seed(3)
    B, n_p, n_reg, D = 3, 10, 2, 32
    N = n_p + 1 + n_reg
    T = dict(patch=torch.randn(B * n_p, D, generator=g), cls=torch.randn(D, generator=g), pos=torch.randn(n_p + 1, D, generator=g), mt=torch.randn(D, generator=g),
             masks=(torch.rand(B, n_p, generator=g) < 0.4).to(torch.uint8), reg=torch.randn(n_reg, D, generator=g), x=torch.zeros(B, N, D),
             dxin=torch.randn(B, N, D, generator=g), dpatch=torch.zeros(B * n_p, D).bfloat16(), dcls=torch.zeros(D), dpos=torch.zeros(n_p + 1, D), dmask=torch.zeros(D),
             dreg=torch.zeros(n_reg, D))
    r = _both(lambda ops, t: (ops.assemble_tokens(t["patch"], t["cls"], t["pos"], t["mt"], t["masks"], B, n_p, D, out=t["x"], reg=t["reg"], n_reg=n_reg),
                              ops.assemble_tokens_bwd(t["dxin"], t["masks"], t["dpatch"], t["dcls"], t["dpos"], t["dmask"], B, n_p, D, dreg=t["dreg"], n_reg=n_reg)),
              T, ["x", "dpatch", "dcls", "dpos", "dmask", "dreg"])
    for k in r:
        _close(r[k], atol=1e-2)


def test_cross_entropy_with_two_targets_slots_and_row_weights():
    g = torch.Generator().manual_seed(4)
    R, K, Tn = 40, 256, 24
    T = dict(s=torch.randn(R, K, generator=g), t=torch.softmax(torch.randn(Tn, K, generator=g), -1), ta=torch.randint(0, Tn, (R,), generator=g, dtype=torch.int32),
             tb=torch.where(torch.rand(R, generator=g) < 0.5, torch.randint(0, Tn, (R,), generator=g, dtype=torch.int32), torch.full((R,), -1, dtype=torch.int32)),
             w=torch.rand(R, generator=g), slot=torch.randint(0, 3, (R,), generator=g, dtype=torch.int32), loss=torch.zeros(5), d=torch.zeros(R, K).bfloat16())
    r = _both(lambda ops, t: ops.ce_fwd_bwd(t["s"], t["t"], t["ta"], t["tb"], t["w"], 0.7, 10.0, t["loss"], t["d"], R, K, slot=t["slot"]), T, ["loss", "d"])
    _close(r["loss"], atol=1e-4); _close(r["d"])


def test_fused_centering_contract():
    """tests/tools/ops_emu.py's statements of lt_softmax_stats_colsum / lt_ce_fwd_bwd_logits against the kernels (two centers, two targets)."""
    g = torch.Generator().manual_seed(14)
    R, K, Ta, Tb = 40, 256, 10, 14
    Tn = Ta + Tb
    T = dict(s=torch.randn(R, K, generator=g), tl=torch.randn(Tn, K, generator=g) * 0.3, ca=torch.randn(K, generator=g) * 0.1, cb=torch.randn(K, generator=g) * 0.1,
             sws=torch.zeros(256 * K), st=torch.zeros(Tn, 2), csa=torch.zeros(K), csb=torch.zeros(K), ta=torch.randint(0, Tn, (R,), generator=g, dtype=torch.int32),
             tb=torch.where(torch.rand(R, generator=g) < 0.5, torch.randint(0, Tn, (R,), generator=g, dtype=torch.int32), torch.full((R,), -1, dtype=torch.int32)),
             w=torch.rand(R, generator=g), slot=torch.randint(0, 3, (R,), generator=g, dtype=torch.int32), loss=torch.zeros(5), d=torch.zeros(R, K).bfloat16())

    def call(ops, t):
        ops.softmax_stats_colsum(t["tl"][:Ta], t["ca"], t["st"][:Ta], t["csa"], Ta, K, 1.0 / 0.05, t["sws"])
        ops.softmax_stats_colsum(t["tl"][Ta:], t["cb"], t["st"][Ta:], t["csb"], Tb, K, 1.0 / 0.05, t["sws"])
        ops.ce_fwd_bwd_logits(t["s"], t["tl"], t["st"], t["ca"], t["cb"], Ta, t["ta"], t["tb"], t["w"], 0.7, 10.0, 1.0 / 0.05, t["loss"], t["d"], R, K, slot=t["slot"])

    r = _both(call, T, ["st", "csa", "csb", "loss", "d"])
    _close(r["st"], atol=1e-5); _close(r["csa"], atol=1e-5); _close(r["csb"], atol=1e-5); _close(r["loss"], atol=1e-4); _close(r["d"])


def test_paka_kernels_contract():
    """lt_roi_resample_tokens (+ its gather-form backward), lt_center_tokens and lt_cka_fwd_bwd against their plain-torch statements: per-image
    4-tap tables with a source-image map, an offset token view (cls row skipped), pad columns of the Gram matrices, an image with coef 0."""
    g = torch.Generator().manual_seed(21)
    B, Bs, n_in, n_out, D, C = 6, 4, 12, 9, 64, 32
    N = n_in + 1
    ld = 16
    x = torch.randn(Bs * N * D, generator=g)
    idx = torch.randint(0, n_in, (B, n_out, 4), generator=g, dtype=torch.int32)
    w = torch.rand(B, n_out, 4, generator=g)
    w[0, 0, 1] = 0.0
    src = torch.randint(0, Bs, (B,), generator=g, dtype=torch.int32)
    z = torch.randn(B * n_out, C, generator=g)
    Ks = torch.randn(B * n_out, ld, generator=g); Kt = torch.randn(B * n_out, ld, generator=g)
    coef = torch.rand(B, generator=g); coef[2] = 0.0
    T = dict(x=x, idx=idx, w=w, src=src, ob=torch.zeros(B * n_out, D).bfloat16(), of=torch.zeros(B * n_out, D), dout=torch.randn(B * n_out, D, generator=g),
             din=torch.zeros(B * N * D), z=z, zc=torch.zeros(B * n_out, C), zb=torch.zeros(B * n_out, C).bfloat16(), Ks=Ks, Kt=Kt, coef=coef, loss=torch.zeros(1),
             G=torch.zeros(B * n_out, ld).bfloat16())

    def call(ops, t):
        ops.roi_resample_tokens(t["x"][D:], t["src"], t["idx"], t["w"], B, N * D, n_out, D, out_bf16=t["ob"], out_f32=t["of"])
        ops.roi_resample_tokens_bwd(t["dout"], t["idx"], t["w"], t["din"][D:], B, N * D, n_in, n_out, D)
        ops.center_tokens(t["z"], B, n_out, C, out_bf16=t["zb"], out_f32=t["zc"])
        ops.cka_fwd_bwd(t["Ks"], t["Kt"], t["coef"], t["loss"], t["G"], B, n_out, ld)

    r = _both(call, T, ["ob", "of", "din", "zc", "zb", "loss", "G"])
    _close(r["of"], atol=1e-5); _close(r["ob"]); _close(r["din"], atol=1e-5); _close(r["zc"], atol=1e-5); _close(r["zb"]); _close(r["loss"], atol=1e-5); _close(r["G"])
    assert float(r["din"][0].view(B, N, D)[:, 0].abs().max()) == 0.0       # the cls rows in front of every image stay untouched


def test_softmax_center_sinkhorn_and_center_ema():
    g = torch.Generator().manual_seed(5)
    R, K = 48, 128
    T = dict(l=torch.randn(R, K, generator=g), c=torch.randn(K, generator=g) * 0.1, p=torch.zeros(R, K), Q=torch.zeros(R, K), cs=torch.zeros(K),
             cen=torch.randn(K, generator=g), colsum=torch.randn(K, generator=g))

    def call(ops, t):
        ops.softmax_center(t["l"], t["c"], t["p"], R, K, 1.0 / 0.05)
        ops.sk_exp(t["l"], t["Q"], 1.0 / 0.3)
        for it in range(3):
            ops.colsum_f32(t["Q"], t["cs"], R, K)
            ops.sk_iter(t["Q"], t["cs"], R, K, float(R), float(R) if it == 2 else 1.0)
        ops.center_ema(t["cen"], t["colsum"], 0.25, 0.9, K)
    r = _both(call, T, ["p", "Q", "cen"])
    for k in r:
        _close(r[k], atol=1e-4)


def test_layerscale_gradient_from_the_weight_gradient_and_row_moves():
    g = torch.Generator().manual_seed(6)
    N, K, R, D = 32, 48, 50, 32
    T = dict(w=(torch.randn(N, K, generator=g) * 0.2).bfloat16(), dw=torch.randn(N, K, generator=g), b=torch.randn(N, generator=g), dbv=torch.randn(N, generator=g),
             gam=torch.randn(N, generator=g) + 2.0, dgam=torch.zeros(N), src=torch.randn(R, D, generator=g), idx=torch.randperm(R, generator=g)[:20],
             ob=torch.zeros(20, D).bfloat16(), of=torch.zeros(20, D), dst=torch.randn(R, D, generator=g), add=torch.randn(20, D, generator=g))
    r = _both(lambda ops, t: (ops.layerscale_dgamma(t["w"], t["dw"], t["b"], t["dbv"], t["gam"], t["dgam"], N, K),
                              ops.gather_rows(t["src"], D, t["idx"], 20, D, out_bf16=t["ob"], out_f32=t["of"]),
                              ops.scatter_add_rows(t["add"], t["idx"], t["dst"], D, 20, D)), T, ["dgam", "ob", "of", "dst"])
    for k in r:
        _close(r[k], atol=1e-2)


def test_optimizer_and_ema_contracts():
    g = torch.Generator().manual_seed(7)
    n = 4096
    T = dict(p=torch.randn(n, generator=g), gr=torch.randn(n, generator=g), m=torch.randn(n, generator=g) * 0.1, v=torch.rand(n, generator=g) * 0.1, pb=torch.zeros(n).bfloat16(),
             soc=torch.tensor([0, 0, 1, 2], dtype=torch.int32), lr=torch.tensor([1e-2, 5e-3, 2e-2]), wd=torch.tensor([1, 0, 1], dtype=torch.uint8),
             fr=torch.tensor([0, 1, 2], dtype=torch.uint8), ss=torch.zeros(1), tea=torch.randn(n, generator=g), tb=torch.zeros(n).bfloat16())

    def call(ops, t):
        ops.sumsq(t["gr"], t["ss"])
        ops.adamw_flat(t["p"], t["gr"], t["m"], t["v"], t["pb"], t["soc"], t["lr"], t["wd"], t["fr"], 1, 0.5, 0.04, 0.9, 0.999, 1e-8, 3, t["ss"], 3.0)
        ops.ema_flat(t["tea"], t["p"], t["tb"], 0.992)
    r = _both(call, T, ["p", "m", "v", "pb", "ss", "tea", "tb"])
    for k in r:
        _close(r[k], atol=1e-5 if k not in ("pb", "tb") else 1e-2)


def test_attention_forward_and_backward():
    g = torch.Generator().manual_seed(8)
    for (B, N, H, dh) in ((3, 37, 2, 64), (2, 70, 1, 64), (2, 21, 2, 16)):
        T = dict(qkv=(torch.randn(B, N, 3 * H * dh, generator=g) * 0.7).bfloat16(), out=torch.zeros(B, N, H * dh).bfloat16(), lse=torch.zeros(B, H, N),
                 dout=torch.randn(B, N, H * dh, generator=g).bfloat16(), dqkv=torch.zeros(B, N, 3 * H * dh).bfloat16())

        def call(ops, t):
            ops.attention_fwd(t["qkv"], t["out"], t["lse"], B, N, H, dh, dh ** -0.5)
            ws = torch.empty(max(8, ops.attention_bwd_ws_floats(B, N, H, dh)), device=t["qkv"].device)
            ops.attention_bwd(t["qkv"], t["out"], t["dout"], t["lse"], ws, t["dqkv"], B, N, H, dh, dh ** -0.5)
        r = _both(call, T, ["out", "dqkv"])
        _close(r["out"]); _close(r["dqkv"], atol=3e-2)


def test_patch_embedding_inputs_and_small_matmul():
    g = torch.Generator().manual_seed(9)
    img = torch.randn(2, 3, 28, 30, generator=g)
    from lightly_train_amd import ops as real_ops
    iy, wy = real_ops.bicubic_taps(28, 32); ix, wx = real_ops.bicubic_taps(30, 32)
    T = dict(img=img, iy=iy, wy=wy, ix=ix, wx=wx, a=torch.randn(20, 12, generator=g), b=torch.randn(12, 16, generator=g), c=torch.randn(20, 16, generator=g),
             w=torch.randn(8, 147, generator=g), wp=torch.zeros(8, 152).bfloat16(), acc=torch.randn(8, 147, generator=g), src=torch.randn(8, 152, generator=g))
    res = {}

    def call(ops, t):
        res[t["img"].device.type] = (ops.resize_4tap(t["img"], t["iy"], t["wy"], t["ix"], t["wx"], 32, 32), )
        res[t["img"].device.type] += (ops.im2col(res[t["img"].device.type][0], 8, 192), )
        ops.matmul_f32(t["a"], t["b"], t["c"], 20, 16, 12, accumulate=True)
        ops.cast_pad_rows(t["w"], t["wp"], 8, 147, 152)
        ops.unpad_accumulate(t["src"], t["acc"], 8, 147, 152)
    r = _both(call, T, ["c", "wp", "acc"])
    for k in r:
        _close(r[k], atol=1e-2)
    _close((res["cuda"][0].float().cpu(), res["cpu"][0].float()), atol=1e-4)
    _close((res["cuda"][1].float().cpu(), res["cpu"][1].float()), atol=1e-2)


def test_head_pieces_batchnorm_and_gelu():
    g = torch.Generator().manual_seed(10)
    R, D, K = 50, 32, 96
    x = torch.randn(R, D, generator=g)
    T = dict(x=x, y=torch.zeros(R, D).bfloat16(), inv=torch.zeros(R), dy=torch.randn(R, D, generator=g), dx=torch.zeros(R, D).bfloat16(),
             v=torch.randn(K, D, generator=g), gg=torch.rand(K, 1, generator=g) + 0.5, w=torch.zeros(K, D).bfloat16(), dw=torch.randn(K, D, generator=g),
             dv=torch.zeros(K, D), dg=torch.zeros(K, 1), cs=torch.zeros(D), xb=torch.randn(R, D, generator=g).bfloat16(),
             u=torch.zeros(R, D).bfloat16(), du=torch.zeros(R, D).bfloat16(), dyb=torch.randn(R, D, generator=g).bfloat16(),
             gam=torch.rand(D, generator=g) + 0.5, bet=torch.randn(D, generator=g), bny=torch.zeros(R, D).bfloat16(), mean=torch.zeros(D), rstd=torch.zeros(D),
             rm=torch.zeros(D), rv=torch.ones(D), bndx=torch.zeros(R, D).bfloat16(), dgam=torch.zeros(D), dbet=torch.zeros(D))

    def call(ops, t):
        ops.l2norm_fwd(t["x"], t["y"], t["inv"], R, D, 1e-12)
        ops.l2norm_bwd(t["dy"], t["x"], t["inv"], t["dx"], R, D)
        ops.weightnorm_fwd(t["v"], t["gg"], t["w"], K, D)
        ops.weightnorm_bwd(t["dw"], t["v"], t["gg"], t["dv"], t["dg"], K, D)
        ops.colsum_bf16(t["xb"], t["cs"], R, D)
        ops.gelu_fwd(t["xb"], t["u"], R * D)
        ops.gelu_bwd(t["dyb"], t["xb"], t["du"], R * D)
        ws = torch.empty(max(8, ops.batchnorm_ws_floats(D)), device=t["x"].device)
        ops.batchnorm_fwd(t["xb"], t["gam"], t["bet"], t["bny"], t["mean"], t["rstd"], R, D, ws, running_mean=t["rm"], running_var=t["rv"])
        ops.batchnorm_bwd(t["dyb"], t["xb"], t["gam"], t["mean"], t["rstd"], t["bndx"], R, D, ws, dgamma=t["dgam"], dbeta=t["dbet"])
    r = _both(call, T, ["y", "inv", "dx", "w", "dv", "dg", "cs", "u", "du", "bny", "mean", "rstd", "rm", "rv", "bndx", "dgam", "dbet"])
    for k in r:
        _close(r[k], atol=2e-2)


def test_distillation_pieces_and_lars():
    g = torch.Generator().manual_seed(11)
    B, n_in, n_out, D, K = 2, 9, 16, 8, 64
    from lightly_train_amd import ops as real_ops
    (idx, w, taps), _ = real_ops.resample_tables(3, 3, 4, 4)
    n = 3072
    T = dict(x=torch.randn(B, n_in, D, generator=g), idx=idx, w=w, out=torch.zeros(B, n_out, D), s=torch.randn(10, K, generator=g), t=torch.randn(10, K, generator=g),
             loss=torch.zeros(1), dl=torch.zeros(10, K).bfloat16(), d=torch.randn(B, 6, 8, generator=g).bfloat16(), gsym=torch.zeros(B, 6, 8).bfloat16(),
             img=torch.randn(4, 3, 4, 4, generator=g), perm=torch.tensor([2, 0, 3, 1]), mix=torch.zeros(4, 3, 4, 4), ms=torch.randn(100, generator=g),
             mt=torch.randn(100, generator=g), mds=torch.zeros(100), mloss=torch.zeros(1), xs=torch.randn(16, 2, 2, 8, generator=g).bfloat16(),
             p=torch.randn(n, generator=g), gr=torch.randn(n, generator=g) * 0.1, buf=torch.randn(n, generator=g) * 0.01, pb=torch.zeros(n).bfloat16(),
             soc=torch.tensor([0, 1, 2], dtype=torch.int32), scb=torch.tensor([0, 1, 2, 3], dtype=torch.int32), lr=torch.tensor([0.5, 0.5, 0.25]),
             wd=torch.tensor([1, 0, 1], dtype=torch.uint8), lws=torch.zeros(8), sn=torch.zeros(3, 2), ss=torch.zeros(1))

    def call(ops, t):
        ops.resample_tokens(t["x"], t["idx"], t["w"], t["out"], B, n_in, n_out, D, taps)
        ops.kl_fwd_bwd(t["s"], t["t"], K, 1.0 / 0.07, 0.3, t["loss"], t["dl"], K, 10, K)
        ops.symmetrize_bf16(t["d"], t["gsym"], B, 6, 8)
        ops.mixup(t["img"], t["perm"], 0.3, t["mix"])
        ops.mse_fwd_bwd(t["ms"], t["mt"], t["mds"], 100, 0.01, t["mloss"])
        ops.sumsq(t["gr"], t["ss"])
        ops.lars_flat(t["p"], t["gr"], t["buf"], t["pb"], t["soc"], t["scb"], t["lr"], t["wd"], t["lws"], t["sn"], 0.5, 1e-3, 0.9, 0.0, False, 0.01, 1e-8, False,
                      t["ss"], 1.0)
    r = _both(call, T, ["out", "loss", "dl", "gsym", "mix", "mds", "mloss", "p", "buf", "pb"])
    for k in r:
        _close(r[k], atol=1e-2 if k in ("dl", "gsym", "pb") else 1e-4)
