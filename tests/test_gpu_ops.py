"""-m gpu: every HIP op of the C ABI against a plain PyTorch fp32 reference of the same op.

bf16 tolerances: operands are rounded to bf16 before both paths, accumulation is fp32 in both, so GEMM-like
ops agree to ~1e-5 relative (summation order); outputs stored as bf16 add 2^-9 relative rounding."""
import math

import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

DEV = "cuda"


def ops():
    import lightly_train_amd  # noqa: F401
    from lightly_train_amd import ops as o

    return o


def bf(x):
    return x.to(torch.bfloat16)


def rel_err(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 768), (197 * 3, 2304, 768), (37, 72, 40), (1, 8, 8), (300, 136, 1000)])
@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, True), (True, False)])
def test_gemm_layouts(M, N, K, ta, tb):
    o = ops()
    if ta and M % 8:
        pytest.skip("transposed A needs M % 8 == 0")
    if tb and N % 8:
        pytest.skip("transposed B needs N % 8 == 0")
    g = torch.Generator(device="cpu").manual_seed(M * 7 + N * 3 + K)
    A = bf(torch.randn(M, K, generator=g)).to(DEV)
    B = bf(torch.randn(N, K, generator=g)).to(DEV)
    ref = A.float() @ B.float().t()
    a_in = A.t().contiguous() if ta else A
    b_in = B.t().contiguous() if tb else B
    out = torch.empty(M, N, device=DEV, dtype=torch.float32)
    o.gemm(a_in, b_in, out, M=M, N=N, K=K, trans_a=ta, trans_b=tb, epilogue=o.EPI_F32)
    naive = o.gemm_naive(a_in, b_in, M, N, K, ta, tb)
    torch.cuda.synchronize()
    assert rel_err(naive, ref) < 1e-5
    assert rel_err(out, ref) < 1e-5, f"mfma gemm mismatch ta={ta} tb={tb}"


@pytest.mark.parametrize("M,N,K,fk", [(m, n, k, f) for (m, n, k) in [(2048, 256, 64), (2500, 768, 768), (4099, 2304, 192), (2048, 264, 128)] for f in (2, 8, 11)] +
                         # automatic dispatch on ViT-S widths (N = 256 m + r: the last column tile is partly padding)
                         [(2309, 384, 384, 0), (4096, 1152, 384, 0), (2048, 328, 1536, 0), (2100, 640, 128, 0)] +
                         # K not a multiple of 64 (SwiGLU widths 2736 / 5472 and small cases): the partial last K-tile of the four-phase kernel
                         [(2100, 1024, 2736, 0), (2048, 512, 5472, 0), (2300, 256, 72, 0), (2048, 304, 200, 8), (2500, 768, 136, 8), (2048, 304, 200, 11), (2500, 768, 136, 11),
                          (2300, 256, 72, 11), (4099, 512, 2736, 11)])
@pytest.mark.parametrize("tb", [False, True])
def test_gemm_256_kernel(M, N, K, tb, fk):
    """the 256x256 LDS-DMA kernel (forced, or dispatched) against the fp32 reference, incl. M/N tails and every
    epilogue it serves."""
    o = ops()
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    A = bf(torch.randn(M, K, generator=g)).to(DEV)
    B = bf(torch.randn(N, K, generator=g) * 0.1).to(DEV)
    ref = A.float() @ B.float().t()
    b_in = B.t().contiguous() if tb else B
    out = torch.empty(M, N, device=DEV, dtype=torch.float32)
    o.gemm(A, b_in, out, M=M, N=N, K=K, trans_b=tb, epilogue=o.EPI_F32, force_kernel=fk)
    assert rel_err(out, ref) < 1e-5
    bias = torch.randn(N, generator=g).to(DEV); gamma = torch.randn(N, generator=g).to(DEV)
    resid = torch.randn(M, N, generator=g).to(DEV); aux = bf(torch.randn(M, N, generator=g)).to(DEV)
    y = ref + bias
    ob = torch.empty(M, N, device=DEV, dtype=torch.bfloat16); pre = torch.empty_like(ob)
    o.gemm(A, b_in, ob, M=M, N=N, K=K, trans_b=tb, epilogue=o.EPI_BF16_GELU, bias=bias, out2=pre, force_kernel=fk)
    assert rel_err(pre, y) < 6e-3 and rel_err(ob, F.gelu(y)) < 6e-3
    y2 = torch.empty_like(ob)
    o.gemm(A, b_in, out, M=M, N=N, K=K, trans_b=tb, epilogue=o.EPI_RESID, bias=bias, gamma=gamma, resid=resid, out2=y2, force_kernel=fk)
    assert rel_err(out, resid + gamma * y) < 1e-5 and rel_err(y2, y) < 6e-3
    o.gemm(A, b_in, ob, M=M, N=N, K=K, trans_b=tb, epilogue=o.EPI_BF16_GELUGRAD, aux=aux, force_kernel=fk)
    x = aux.float().requires_grad_(True)
    F.gelu(x).sum().backward()
    assert rel_err(ob, ref * x.grad) < 6e-3


@pytest.mark.parametrize("M,N,K,fk", [(m, n, k, f) for (m, n, k) in [(768, 768, 8192), (3072, 768, 12800), (256, 136, 8256), (512, 768, 8256)] for f in (2, 8, 11)] +
                         # automatic dispatch: few output rows (ResNet layer1 / layer2 convolutions) and a layer4-sized contraction
                         [(64, 576, 16384, 0), (128, 1152, 8192, 0), (512, 4608, 6272, 0), (2048, 512, 4096, 0)])
def test_gemm_256_wgrad_slab_and_atomic(M, N, K, fk):
    """dW[M,N] += dY[K,M]^T X[K,N] on the 256-row kernel (forced, or as the dispatcher picks it for weight gradients with >= 64 output
    rows over >= 4096 contraction rows): deterministic slab split-K and the atomic variant."""
    o = ops()
    g = torch.Generator().manual_seed(K)
    dY = bf(torch.randn(K, M, generator=g)).to(DEV)
    X = bf(torch.randn(K, N, generator=g)).to(DEV)
    ref = 1.0 + 0.5 * (dY.float().t() @ X.float())
    slab = torch.empty(64 * 1024 * 1024, device=DEV)
    outs = []
    for ws in (slab, None, slab):
        dW = torch.ones(M, N, device=DEV)
        o.gemm(dY, X, dW, M=M, N=N, K=K, trans_a=True, trans_b=True, epilogue=o.EPI_F32_ACCUM, split_k=2, alpha=0.5, lda=M, ldb=N,
               force_kernel=fk, workspace=ws)
        assert rel_err(dW, ref) < 2e-5
        outs.append(dW)
    assert torch.equal(outs[0], outs[2]), "slab split-K must be bit-reproducible"


@pytest.mark.parametrize("M,N,K,fk", [(2304, 768, 50432, 0), (3072, 768, 51200, 8), (3072, 768, 51200, 11), (768, 768, 50176, 0), (256, 2048, 8832, 8), (256, 2048, 8832, 11), (2048, 768, 8832, 0),
                                      (2000, 776, 8192, 8), (64, 576, 16384, 0), (768, 768, 1000, 0)])
def test_gemm_wgrad_fused_bias_column_sums(M, N, K, fk):
    """`colsum=`: db[M] += column sums of dY[K,M] beside dW += dY^T X -- fused into the four-phase slab kernel while the reduction ledger
    is open (partial rows per (slice, column tile, wave column), order-fixed flush), stand-alone launch on every other path.  Checked
    against torch in fp64, bitwise against a second run, and equal (to rounding) to the stand-alone column sum."""
    o = ops()
    g = torch.Generator().manual_seed(M + K)
    dY = bf(torch.randn(K, M, generator=g) + 0.25).to(DEV)     # non-zero mean: the sums do not cancel
    X = bf(torch.randn(K, N, generator=g)).to(DEV)
    ref_w = dY.double().t() @ X.double()
    ref_b = 3.0 + dY.double().sum(0)
    slab = torch.empty(64 * 1024 * 1024, device=DEV)
    scratch = torch.empty(8 * 1024 * 1024, device=DEV)
    outs = []
    for ledger in (True, True, False):
        dW = torch.zeros(M, N, device=DEV)
        db = torch.full((M,), 3.0, device=DEV)
        ovf = o.reduce_overflows()
        if ledger:
            o.reduce_begin(scratch)
        o.gemm(dY, X, dW, M=M, N=N, K=K, trans_a=True, trans_b=True, epilogue=o.EPI_F32_ACCUM, split_k=2, lda=M, ldb=N, force_kernel=fk,
               workspace=slab, colsum=db)
        if ledger:
            o.reduce_end()
        assert o.reduce_overflows() == ovf
        assert rel_err(dW, ref_w) < 2e-5
        assert rel_err(db, ref_b) < 2e-5, (ledger, rel_err(db, ref_b))
        outs.append((dW, db))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), "fused column sums must be bit-reproducible"
    db2 = torch.full((M,), 3.0, device=DEV)
    o.colsum_bf16(dY, db2, K, M)
    assert rel_err(outs[0][1], db2) < 1e-5


def test_gemm_asymmetric_identity():
    """A = I with asymmetric B catches row/col swaps in the C write (guide G9)."""
    o = ops()
    n = 128
    A = bf(torch.eye(n)).to(DEV)
    Bm = bf((torch.arange(n)[:, None] * 2 + torch.arange(n)[None, :] % 7).float() / 16).to(DEV)  # B[n][k]
    out = torch.empty(n, n, device=DEV, dtype=torch.float32)
    o.gemm(A, Bm, out, M=n, N=n, K=n, epilogue=o.EPI_F32)
    assert torch.equal(out, Bm.float().t())


def test_gemm_epilogues():
    o = ops()
    M, N, K = 200, 256, 192
    g = torch.Generator().manual_seed(0)
    A = bf(torch.randn(M, K, generator=g)).to(DEV)
    W = bf(torch.randn(N, K, generator=g) * 0.05).to(DEV)
    bias = torch.randn(N, generator=g).to(DEV)
    gamma = torch.randn(N, generator=g).to(DEV)
    resid = torch.randn(M, N, generator=g).to(DEV)
    y = A.float() @ W.float().t() + bias
    # BF16
    out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    o.gemm(A, W, out, M=M, N=N, K=K, epilogue=o.EPI_BF16, bias=bias)
    assert rel_err(out, y) < 6e-3
    # GELU + pre
    pre = torch.empty_like(out)
    o.gemm(A, W, out, M=M, N=N, K=K, epilogue=o.EPI_BF16_GELU, bias=bias, out2=pre)
    assert rel_err(pre, y) < 6e-3
    assert rel_err(out, F.gelu(y)) < 6e-3
    # RESID
    outf = torch.empty(M, N, device=DEV, dtype=torch.float32)
    ybf = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    o.gemm(A, W, outf, M=M, N=N, K=K, epilogue=o.EPI_RESID, bias=bias, gamma=gamma, resid=resid, out2=ybf)
    assert rel_err(outf, resid + gamma * y) < 1e-5
    assert rel_err(ybf, y) < 6e-3
    # GELUGRAD
    aux = bf(torch.randn(M, N, generator=g)).to(DEV)
    o.gemm(A, W, out, M=M, N=N, K=K, epilogue=o.EPI_BF16_GELUGRAD, aux=aux)
    x = aux.float().requires_grad_(True)
    F.gelu(x).sum().backward()
    assert rel_err(out, (A.float() @ W.float().t()) * x.grad) < 6e-3
    # ACCUM with split-k (atomics) and alpha
    acc = torch.ones(M, N, device=DEV, dtype=torch.float32)
    o.gemm(A, W, acc, M=M, N=N, K=K, epilogue=o.EPI_F32_ACCUM, split_k=3, alpha=0.5)
    assert rel_err(acc, 1 + 0.5 * (A.float() @ W.float().t())) < 1e-5


def test_gemm_wgrad_shape():
    """dW[N,K] += dY[M,N]^T X[M,K] with a long token dimension and split-k."""
    o = ops()
    M, N, K = 3000, 256, 128
    g = torch.Generator().manual_seed(1)
    dY = bf(torch.randn(M, N, generator=g)).to(DEV)
    X = bf(torch.randn(M, K, generator=g)).to(DEV)
    dW = torch.zeros(N, K, device=DEV, dtype=torch.float32)
    o.gemm(dY, X, dW, M=N, N=K, K=M, trans_a=True, trans_b=True, epilogue=o.EPI_F32_ACCUM, split_k=4, lda=N, ldb=K)
    assert rel_err(dW, dY.float().t() @ X.float()) < 1e-5


@pytest.mark.parametrize("rows,D", [(1000, 768), (37, 384), (50, 8), (129, 1024), (5, 64)])
def test_layernorm(rows, D):
    o = ops()
    g = torch.Generator().manual_seed(rows + D)
    x = (torch.randn(rows, D, generator=g) * 2 + 0.5).to(DEV)
    w = (torch.randn(D, generator=g) * 0.2 + 1).to(DEV)
    b = (torch.randn(D, generator=g) * 0.1).to(DEV)
    yb = torch.empty(rows, D, device=DEV, dtype=torch.bfloat16)
    yf = torch.empty(rows, D, device=DEV)
    mean = torch.empty(rows, device=DEV)
    rstd = torch.empty(rows, device=DEV)
    o.layernorm_fwd(x, w, b, rows, D, y_bf16=yb, y_f32=yf, mean=mean, rstd=rstd)
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True)
    ref = F.layer_norm(xr, (D,), wr, br, 1e-6)
    assert (yf - ref).abs().max().item() < 2e-5
    assert rel_err(yb, ref) < 5e-3
    # backward (bf16 dy and f32 dy)
    dy = torch.randn(rows, D, generator=g).to(DEV)
    dres = torch.randn(rows, D, generator=g).to(DEV)
    wsb = torch.empty(2048 * 2 * D, device=DEV)
    for dyt, wsx in ((dy, None), (bf(dy), None), (bf(dy), wsb)):
        dx = torch.empty(rows, D, device=DEV)
        dw = torch.zeros(D, device=DEV)
        db = torch.zeros(D, device=DEV)
        o.layernorm_bwd(x, w, mean, rstd, dyt, dres, dx, dw, db, rows, D, ws=wsx)
        for t in (xr, wr, br):
            t.grad = None
        ref.backward(dyt.float(), retain_graph=True)
        assert rel_err(dx, xr.grad + dres) < 2e-5
        assert rel_err(dw, wr.grad) < 2e-5
        assert rel_err(db, br.grad) < 2e-5


def test_layernorm_bwd_fused_next_branch_and_layerscale_dgamma():
    """(1) LayerNorm backward that also emits the next branch's upstream gradient dx*gamma*rowscale (+ its column sums);
    (2) the LayerScale gradient from the weight gradient: (rowdot(W, dW) + b*db)/gamma == sum_r dx * y."""
    o = ops()
    g = torch.Generator().manual_seed(21)
    rows, D, K = 1000, 768, 256
    x = torch.randn(rows, D, generator=g).to(DEV); w = torch.randn(D, generator=g).to(DEV)
    dy = bf(torch.randn(rows, D, generator=g)).to(DEV); dres = torch.randn(rows, D, generator=g).to(DEV)
    mean = x.mean(1); rstd = (x.var(1, unbiased=False) + 1e-6).rsqrt()
    gam = (torch.randn(D, generator=g) * 1e-3).to(DEV); rsc = (torch.rand(rows, generator=g) < 0.8).float().to(DEV) / 0.8
    dx0, dx1 = torch.empty_like(x), torch.empty_like(x)
    dw0, db0, dw1, db1 = (torch.zeros(D, device=DEV) for _ in range(4))
    o.layernorm_bwd(x, w, mean, rstd, dy, dres, dx0, dw0, db0, rows, D)
    dn = torch.empty(rows, D, device=DEV, dtype=torch.bfloat16); dbn = torch.zeros(D, device=DEV)
    o.layernorm_bwd(x, w, mean, rstd, dy, dres, dx1, dw1, db1, rows, D, dnext=dn, gamma_next=gam, rowscale_next=rsc, scale_next=1.25, dbias_next=dbn)
    assert torch.equal(dx0, dx1) and rel_err(dw1, dw0) < 1e-5 and rel_err(db1, db0) < 1e-5
    ref_dn = dx0 * gam * rsc[:, None] * 1.25
    assert rel_err(dn, ref_dn) < 6e-3 and rel_err(dbn, ref_dn.sum(0)) < 1e-4

    # LayerScale gradient identity on a Linear y = A W^T + b with upstream dD = bf16(dx * gamma)
    A = bf(torch.randn(rows, K, generator=g)).to(DEV); W = bf(torch.randn(D, K, generator=g) * 0.05).to(DEV)
    b = torch.randn(D, generator=g).to(DEV)
    dxu = torch.randn(rows, D, generator=g).to(DEV)
    dD = bf(dxu * gam)
    y = A.float() @ W.float().t() + b
    dW = dD.float().t() @ A.float(); dbias = dD.float().sum(0)
    dgam = torch.zeros(D, device=DEV)
    o.layerscale_dgamma(W, dW.contiguous(), b, dbias, gam, dgam, D, K)
    assert rel_err(dgam, (dxu * y).sum(0)) < 1e-2   # bf16 rounding of dD only


@pytest.mark.parametrize("depth,N,K,with_bias", [(12, 768, 768, True), (2, 384, 1536, True), (5, 256, 64, False)])
def test_layerscale_dgamma_batched_equals_the_per_layer_launches(depth, N, K, with_bias):
    """lt_layerscale_dgamma_batched (every block's LayerScale gradient in one launch at the tail of the step): the arguments are layer 0's
    tensors, layer i's lie i * stride elements further on in the bf16 shadow, the parameter and the gradient storages alike.  Against the
    per-layer launches on the same strided layout: bit for bit (the same arithmetic per row), and accumulating (dgamma += ...)."""
    o = ops()
    g = torch.Generator().manual_seed(depth * 1000 + N + K)
    per = N * K + 2 * N + 37 * 8          # weight | bias | gamma | unrelated tensors of the block: a stride that is not a multiple of the sizes
    per = (per + 7) // 8 * 8
    data = (torch.randn(depth * per, generator=g) * 0.05).to(DEV)
    grad = torch.randn(depth * per, generator=g).to(DEV)
    shadow = data.to(torch.bfloat16)
    data[N * K + N:N * K + 2 * N].add_(1.0)   # gamma away from zero (every layer below gets its own values anyway)
    for i in range(depth):
        data[i * per + N * K + N:i * per + N * K + 2 * N] = (torch.rand(N, generator=g) + 0.5).to(DEV)

    def views(buf, i):
        o0 = i * per
        return buf[o0:o0 + N * K].view(N, K), buf[o0 + N * K:o0 + N * K + N], buf[o0 + N * K + N:o0 + N * K + 2 * N]

    g_ref, g_bat = grad.clone(), grad.clone()
    for i in range(depth):
        wb, _, _ = views(shadow, i)
        _, b, gam = views(data, i)
        dw, db, dgam = views(g_ref, i)
        o.layerscale_dgamma(wb, dw, b if with_bias else None, db if with_bias else None, gam, dgam, N, K)
    wb0, _, _ = views(shadow, 0)
    _, b0, gam0 = views(data, 0)
    dw0, db0, dgam0 = views(g_bat, 0)
    o.layerscale_dgamma_batched(wb0, dw0, b0 if with_bias else None, db0 if with_bias else None, gam0, dgam0, N, K, depth, per)
    torch.cuda.synchronize()
    assert torch.equal(g_ref, g_bat)
    assert not torch.equal(g_bat, grad)          # something was accumulated
    i = depth - 1                                # and it is the LayerScale identity: dgamma += (rowdot(W, dW) + b * db) / gamma
    wb, _, _ = views(shadow, i); _, b, gam = views(data, i); dw, db, _ = views(grad, i)
    want = views(grad, i)[2] + ((wb.float() * dw).sum(1) + (b * db if with_bias else 0.0)) / gam
    assert rel_err(views(g_bat, i)[2], want) < 1e-5


def test_im2col_and_tokens():
    o = ops()
    B, C, H, W, p, D = 3, 3, 32, 48, 16, 24
    g = torch.Generator().manual_seed(2)
    img = torch.randn(B, C, H, W, generator=g).to(DEV)
    cols = o.im2col(img, p, C * p * p + 8)
    ref = F.unfold(img, kernel_size=p, stride=p).transpose(1, 2).reshape(B * (H // p) * (W // p), C * p * p)
    assert torch.equal(cols[:, : C * p * p].float(), bf(ref).float())
    assert cols[:, C * p * p:].abs().max().item() == 0
    n_p = (H // p) * (W // p)
    patch = torch.randn(B * n_p, D, generator=g).to(DEV)
    cls = torch.randn(D, generator=g).to(DEV)
    pos = torch.randn(n_p + 1, D, generator=g).to(DEV)
    mt = torch.randn(D, generator=g).to(DEV)
    masks = (torch.rand(B, n_p, generator=g) < 0.4).to(DEV)
    x = o.assemble_tokens(patch, cls, pos, mt, masks.to(torch.uint8), B, n_p, D)
    t = torch.where(masks.unsqueeze(-1), mt.view(1, 1, D), patch.view(B, n_p, D))
    refx = torch.cat([cls.view(1, 1, D).expand(B, -1, -1), t], 1) + pos.unsqueeze(0)
    assert torch.allclose(x, refx)
    dx = torch.randn(B, n_p + 1, D, generator=g).to(DEV)
    dpatch = torch.empty(B * n_p, D, device=DEV, dtype=torch.bfloat16)
    dcls = torch.zeros(D, device=DEV); dpos = torch.zeros(n_p + 1, D, device=DEV); dmask = torch.zeros(D, device=DEV)
    o.assemble_tokens_bwd(dx, masks.to(torch.uint8), dpatch, dcls, dpos, dmask, B, n_p, D)
    assert torch.allclose(dcls, dx[:, 0].sum(0), atol=1e-5)
    assert torch.allclose(dpos, dx.sum(0), atol=1e-5)
    assert torch.allclose(dmask, (dx[:, 1:] * masks.unsqueeze(-1)).sum((0, 1)), atol=1e-5)
    assert torch.equal(dpatch.float().view(B, n_p, D), bf(dx[:, 1:] * (~masks).unsqueeze(-1)).float())


def test_tokens_with_registers_and_swiglu():
    """Register tokens are inserted after cls without a positional embedding (vision_transformer.py:318-327);
    SwiGLU gate silu(x1)*x2 and its backward (swiglu_ffn.py:31-35) against torch autograd on the bf16-rounded inputs."""
    o = ops()
    B, n_p, R, D = 3, 6, 4, 24
    g = torch.Generator().manual_seed(12)
    patch = torch.randn(B * n_p, D, generator=g).to(DEV)
    cls, mt = torch.randn(D, generator=g).to(DEV), torch.randn(D, generator=g).to(DEV)
    pos, reg = torch.randn(n_p + 1, D, generator=g).to(DEV), torch.randn(R, D, generator=g).to(DEV)
    masks = (torch.rand(B, n_p, generator=g) < 0.4).to(DEV)
    x = o.assemble_tokens(patch, cls, pos, mt, masks.to(torch.uint8), B, n_p, D, reg=reg, n_reg=R)
    t = torch.where(masks.unsqueeze(-1), mt.view(1, 1, D), patch.view(B, n_p, D))
    t = torch.cat([cls.view(1, 1, D).expand(B, -1, -1), t], 1) + pos.unsqueeze(0)
    refx = torch.cat([t[:, :1], reg.unsqueeze(0).expand(B, -1, -1), t[:, 1:]], 1)
    assert x.shape == (B, n_p + 1 + R, D) and torch.allclose(x, refx)
    dx = torch.randn(B, n_p + 1 + R, D, generator=g).to(DEV)
    dpatch = torch.empty(B * n_p, D, device=DEV, dtype=torch.bfloat16)
    dcls, dmask = torch.zeros(D, device=DEV), torch.zeros(D, device=DEV)
    dpos, dreg = torch.zeros(n_p + 1, D, device=DEV), torch.ones(R, D, device=DEV)   # dreg accumulates onto 1
    o.assemble_tokens_bwd(dx, masks.to(torch.uint8), dpatch, dcls, dpos, dmask, B, n_p, D, dreg=dreg, n_reg=R)
    dxp = dx[:, 1 + R:]
    assert torch.allclose(dcls, dx[:, 0].sum(0), atol=1e-5) and torch.allclose(dreg, 1 + dx[:, 1:1 + R].sum(0), atol=1e-5)
    assert torch.allclose(dpos, torch.cat([dx[:, :1], dxp], 1).sum(0), atol=1e-5)
    assert torch.allclose(dmask, (dxp * masks.unsqueeze(-1)).sum((0, 1)), atol=1e-5)
    assert torch.equal(dpatch.float().view(B, n_p, D), bf(dxp * (~masks).unsqueeze(-1)).float())

    rows, H = 37, 176
    x12 = bf(torch.randn(rows, 2 * H, generator=g) * 2).to(DEV)
    dh = bf(torch.randn(rows, H, generator=g)).to(DEV)
    out = torch.empty(rows, H, device=DEV, dtype=torch.bfloat16)
    d12 = torch.empty(rows, 2 * H, device=DEV, dtype=torch.bfloat16)
    o.swiglu_fwd(x12, out, rows, H)
    o.swiglu_bwd(x12, dh, d12, rows, H)
    xr = x12.float().requires_grad_(True)
    ref = F.silu(xr[:, :H]) * xr[:, H:]
    ref.backward(dh.float())
    assert (out.float() - ref.detach()).abs().max().item() <= 1e-2 * ref.abs().max().item()
    assert (d12.float() - xr.grad).abs().max().item() <= 1e-2 * xr.grad.abs().max().item()


def test_distillation_ops_kl_symmetrize_mixup_batched_gemm():
    """Kernels of the DistillationV3 step: row KL with gradient, dS + dS^T, mixup, batched GEMM (per-image X X^T)."""
    o = ops()
    g = torch.Generator().manual_seed(31)
    rows, K, ld, T = 50, 196, 200, 0.07
    s = torch.zeros(rows, ld); t = torch.zeros(rows, ld)
    s[:, :K] = torch.randn(rows, K, generator=g) * 0.3; t[:, :K] = torch.randn(rows, K, generator=g) * 0.3
    sd, td = s.to(DEV), t.to(DEV)
    loss = torch.zeros(1, device=DEV); dl = torch.zeros(rows, ld, device=DEV, dtype=torch.bfloat16)
    o.kl_fwd_bwd(sd, td, ld, 1.0 / T, 1.0 / rows, loss, dl, ld, rows, K)
    sr = s[:, :K].clone().requires_grad_(True)
    ref = torch.nn.KLDivLoss(reduction="batchmean")(F.log_softmax(sr / T, -1), F.softmax(t[:, :K] / T, -1))
    ref.backward()
    assert float(loss) == pytest.approx(float(ref), rel=1e-4)
    assert rel_err(dl[:, :K].cpu(), sr.grad) < 6e-3 and dl[:, K:].abs().max().item() == 0

    B, n = 3, 20
    d = bf(torch.randn(B, n, 24, generator=g)).to(DEV)
    gsym = torch.zeros_like(d)
    o.symmetrize_bf16(d, gsym, B, n, 24)
    assert rel_err(gsym[:, :, :n].cpu(), (d[:, :, :n].float() + d[:, :, :n].float().transpose(1, 2)).cpu()) < 6e-3

    x = torch.randn(5, 3, 8, 8, generator=g).to(DEV); idx = torch.randperm(5, generator=g)
    out = torch.empty_like(x)
    o.mixup(x, idx.to(DEV), 0.3, out)
    assert torch.allclose(out, 0.3 * x + 0.7 * x[idx.to(DEV)], atol=1e-6)

    Bq, m, Kd = 4, 196, 64                                  # per-image token similarity S_b = X_b X_b^T into a padded [m, 200] tile
    X = bf(torch.randn(Bq * m, Kd, generator=g) * 0.2).to(DEV)
    S = torch.zeros(Bq * m, 200, device=DEV)
    o.gemm(X, X, S, M=m, N=m, K=Kd, epilogue=o.EPI_F32, ldc=200, batch=Bq, stride_a=m * Kd, stride_b=m * Kd, stride_c=m * 200)
    Xf = X.float().view(Bq, m, Kd)
    assert rel_err(S.view(Bq, m, 200)[:, :, :m].cpu(), (Xf @ Xf.transpose(1, 2)).cpu()) < 1e-5 and S.view(Bq, m, 200)[:, :, m:].abs().max().item() == 0


def test_bicubic_pad_resize():
    """98x98 -> 112x112 (the literal 8 x 98^2 local crops with patch 16): taps read off F.interpolate, applied in HIP."""
    o = ops()
    g = torch.Generator().manual_seed(8)
    img = torch.randn(3, 3, 98, 90, generator=g).to(DEV)
    iy, wy = o.bicubic_taps(98, 112)
    ix, wx = o.bicubic_taps(90, 96)
    out = o.resize_4tap(img, iy.to(DEV), wy.to(DEV), ix.to(DEV), wx.to(DEV), 112, 96)
    ref = F.interpolate(img.cpu(), size=(112, 96), mode="bicubic", align_corners=False)
    assert (out.cpu() - ref).abs().max().item() < 2e-5


def test_layerscale_colsum_gather():
    o = ops()
    rows, D = 333, 136
    g = torch.Generator().manual_seed(3)
    dout = torch.randn(rows, D, generator=g).to(DEV)
    y = bf(torch.randn(rows, D, generator=g)).to(DEV)
    gamma = torch.randn(D, generator=g).to(DEV)
    dy = torch.empty(rows, D, device=DEV, dtype=torch.bfloat16)
    dg = torch.zeros(D, device=DEV); dbias = torch.zeros(D, device=DEV)
    o.layerscale_bwd(dout, y, gamma, dy, dg, rows, D, dbias=dbias)
    assert torch.equal(dy.float(), bf(dout * gamma).float())
    assert rel_err(dg, (dout * y.float()).sum(0)) < 1e-5
    assert rel_err(dbias, (dout * gamma).sum(0)) < 1e-5
    dy2 = torch.empty(rows, 10, device=DEV, dtype=torch.bfloat16)  # scalar fallback, no gamma
    o.layerscale_bwd(dout[:, :10].contiguous(), None, None, dy2, None, rows, 10)
    assert torch.equal(dy2.float(), bf(dout[:, :10]).float())
    cs = torch.zeros(D, device=DEV)
    o.colsum_bf16(y, cs, rows, D)
    assert rel_err(cs, y.float().sum(0)) < 1e-5
    cs2 = torch.empty(D, device=DEV)
    o.colsum_f32(dout, cs2, rows, D)
    assert rel_err(cs2, dout.sum(0)) < 1e-5
    idx = torch.randperm(rows, generator=g)[:57].to(DEV)
    gb = torch.empty(57, D, device=DEV, dtype=torch.bfloat16); gf = torch.empty(57, D, device=DEV)
    o.gather_rows(dout, D, idx, 57, D, out_bf16=gb, out_f32=gf)
    assert torch.equal(gf, dout[idx]) and torch.equal(gb.float(), bf(dout[idx]).float())
    dst = torch.zeros(rows, D, device=DEV)
    o.scatter_add_rows(gf, idx, dst, D, 57, D)
    ref = torch.zeros(rows, D, device=DEV); ref[idx] = gf
    assert torch.equal(dst, ref)


def _attn_ref(qkv, B, N, H, dh, scale, dout=None):
    q, k, v = qkv.float().view(B, N, 3, H, dh).permute(2, 0, 3, 1, 4)
    q = q.detach().requires_grad_(True); k = k.detach().requires_grad_(True); v = v.detach().requires_grad_(True)
    s = (q * scale) @ k.transpose(-2, -1)
    a = s.softmax(-1)
    out = (a @ v).transpose(1, 2).reshape(B, N, H * dh)
    lse = torch.logsumexp(s, -1)
    if dout is None:
        return out, lse, None
    out.backward(dout.float().view(B, N, H * dh))
    dqkv = torch.stack([q.grad, k.grad, v.grad], 0).permute(1, 3, 0, 2, 4).reshape(B, N, 3 * H * dh)
    return out, lse, dqkv


@pytest.mark.parametrize("B,N,H,dh", [(2, 197, 3, 64), (3, 37, 2, 64), (2, 50, 1, 64), (1, 300, 2, 64), (2, 128, 2, 64),
                                       (3, 50, 6, 64), (2, 64, 4, 64), (2, 257, 2, 64), (1, 600, 1, 64), (2, 65, 2, 64),
                                       (4, 17, 2, 4), (2, 5, 2, 8),
                                       (1, 1370, 2, 64), (2, 1374, 1, 64),    # BASELINE cfg5: ViT-L/14 at 518^2 = 37x37 + cls (+ 4 registers)
                                       # fused backward (65..224 tokens): tile / chunk edges, 4 registers, the 224-token cap and one past it
                                       (2, 201, 3, 64), (1, 224, 2, 64), (2, 225, 1, 64), (2, 193, 2, 64), (3, 96, 2, 64), (2, 100, 4, 64),
                                       (130, 197, 2, 64),
                                       # packed block-diagonal backward (32..64 tokens: several images per 224-token sequence, ragged last sequence)
                                       (9, 37, 2, 64), (5, 50, 2, 64), (7, 33, 2, 64), (5, 32, 2, 64), (4, 48, 3, 64), (33, 50, 4, 64), (7, 64, 2, 64),
                                       # fused two-heads-per-block backward (N <= 64, even head count): one key tile only, a ragged first tile, 12 heads
                                       (2, 20, 2, 64), (3, 31, 4, 64), (2, 2, 2, 64), (2, 50, 12, 64),
                                       # fused backward in two key passes (225..288 tokens: the patch-14 global crops): eight and nine tiles, ragged last tile, the cap, head walk
                                       (2, 261, 3, 64), (1, 288, 2, 64), (3, 256, 2, 64), (2, 240, 1, 64), (44, 257, 12, 64), (2, 289, 2, 64)])
def test_attention(B, N, H, dh):
    o = ops()
    g = torch.Generator().manual_seed(N + dh)
    qkv = bf(torch.randn(B, N, 3 * H * dh, generator=g)).to(DEV)
    scale = dh ** -0.5
    out = torch.zeros(B, N, H * dh, device=DEV, dtype=torch.bfloat16)
    lse = torch.zeros(B, H, N, device=DEV)
    o.attention_fwd(qkv, out, lse, B, N, H, dh, scale)
    dout = bf(torch.randn(B, N, H * dh, generator=g)).to(DEV)
    ref_out, ref_lse, ref_dqkv = _attn_ref(qkv, B, N, H, dh, scale, dout)
    assert rel_err(lse, ref_lse) < 1e-4
    assert rel_err(out, ref_out) < 1.5e-2, "attention forward"
    ws = torch.zeros(o.attention_bwd_ws_floats(B, N, H, dh), device=DEV)
    dqkv = torch.zeros(B, N, 3 * H * dh, device=DEV, dtype=torch.bfloat16)
    o.attention_bwd(qkv, out, dout, lse, ws, dqkv, B, N, H, dh, scale)
    d = dqkv.float().view(B, N, 3, H * dh)
    r = ref_dqkv.view(B, N, 3, H * dh)
    for i, nm in enumerate("qkv"):
        assert rel_err(d[:, :, i], r[:, :, i]) < 2.5e-2, f"attention backward d{nm}"


def test_attention_rescale_branch():
    """spiked key forces the online-softmax running max to jump at a late tile (guide rule 26)."""
    o = ops()
    B, N, H, dh = 1, 160, 1, 64
    g = torch.Generator().manual_seed(9)
    x = torch.randn(B, N, 3, H, dh, generator=g) * 0.5
    x[0, 150, 1] = x[0, 3, 0] * 8  # key 150 aligned with query 3
    qkv = bf(x.reshape(B, N, -1)).to(DEV)
    out = torch.zeros(B, N, H * dh, device=DEV, dtype=torch.bfloat16); lse = torch.zeros(B, H, N, device=DEV)
    o.attention_fwd(qkv, out, lse, B, N, H, dh, dh ** -0.5)
    ref_out, ref_lse, _ = _attn_ref(qkv, B, N, H, dh, dh ** -0.5)
    assert rel_err(lse, ref_lse) < 1e-4 and rel_err(out, ref_out) < 1.5e-2


def test_head_pieces():
    o = ops()
    rows, D, K = 77, 256, 1024
    g = torch.Generator().manual_seed(4)
    x = torch.randn(rows, D, generator=g).to(DEV)
    y = torch.empty(rows, D, device=DEV, dtype=torch.bfloat16); inv = torch.empty(rows, device=DEV)
    o.l2norm_fwd(x, y, inv, rows, D)
    xr = x.clone().requires_grad_(True)
    ref = F.normalize(xr, dim=-1, eps=1e-12)
    assert rel_err(y, ref) < 5e-3
    dy = torch.randn(rows, D, generator=g).to(DEV)
    dx = torch.empty(rows, D, device=DEV, dtype=torch.bfloat16)
    o.l2norm_bwd(dy, x, inv, dx, rows, D)
    ref.backward(dy)
    assert rel_err(dx, xr.grad) < 5e-3
    v = torch.randn(K, D, generator=g).to(DEV); gg = (torch.rand(K, 1, generator=g) + 0.5).to(DEV)
    w = torch.empty(K, D, device=DEV, dtype=torch.bfloat16)
    o.weightnorm_fwd(v, gg, w, K, D)
    vr = v.clone().requires_grad_(True); gr = gg.clone().requires_grad_(True)
    wref = torch._weight_norm(vr, gr, 0)
    assert rel_err(w, wref) < 5e-3
    dw = torch.randn(K, D, generator=g).to(DEV)
    dv = torch.zeros(K, D, device=DEV); dg = torch.zeros(K, 1, device=DEV)
    o.weightnorm_bwd(dw, v, gg, dv, dg, K, D)
    wref.backward(dw)
    assert rel_err(dv, vr.grad) < 1e-5 and rel_err(dg, gr.grad) < 1e-5


@pytest.mark.parametrize("T,R,D", [(1000, 640, 768), (300, 37, 384), (513, 513, 1024), (70, 1, 64)])
def test_indexed_row_forms_of_the_subset_branch_backward(T, R, D):
    """Round 5 (batch-subset stochastic depth, block.py:118-141, and the last block's loss rows): lt_layerscale_bwd_rows reads the upstream
    gradient of the R subset rows through their row index instead of a gathered copy, lt_layernorm_bwd_rows adds the LayerNorm backward
    of those rows in place to the rows of the full gradient stream instead of a compact result + scatter-add pass.  Both against the
    two-kernel forms they replace (bitwise for the LayerScale form; the LayerNorm form to fp32 round-off: one add order), rows outside
    the subset untouched."""
    o = ops()
    g = torch.Generator().manual_seed(T + R)
    idx = torch.randperm(T, generator=g)[:R].sort().values.to(torch.int64).to(DEV)
    dx = torch.randn(T, D, generator=g).to(DEV)
    gamma, rowscale = (torch.rand(D, generator=g) + 0.5).to(DEV), (torch.rand(R, generator=g) + 0.5).to(DEV)
    # LayerScale backward: gathered copy vs index
    dxs = torch.empty(R, D, device=DEV)
    o.gather_rows(dx, D, idx, R, D, out_f32=dxs)
    dy0, dy1 = torch.zeros(R, D, device=DEV, dtype=torch.bfloat16), torch.zeros(R, D, device=DEV, dtype=torch.bfloat16)
    db0, db1 = torch.zeros(D, device=DEV), torch.zeros(D, device=DEV)
    o.layerscale_bwd(dxs, None, gamma, dy0, None, R, D, dbias=db0, rowscale=rowscale, scale=1.25)
    o.layerscale_bwd(dx, None, gamma, dy1, None, R, D, dbias=db1, rowscale=rowscale, scale=1.25, ridx=idx)
    assert torch.equal(dy0, dy1) and rel_err(db1, db0) < 1e-5
    # LayerNorm backward: compact result + scatter-add vs in place through the index
    x = torch.randn(R, D, generator=g).to(DEV)
    w = (torch.rand(D, generator=g) + 0.5).to(DEV)
    mean, rstd = x.mean(-1), (x.var(-1, unbiased=False) + 1e-6).rsqrt()
    dyl = bf(torch.randn(R, D, generator=g)).to(DEV)
    lng = torch.empty(R, D, device=DEV)
    dw0, dbn0, dw1, dbn1 = (torch.zeros(D, device=DEV) for _ in range(4))
    ref = dx.clone()
    o.layernorm_bwd(x, w, mean, rstd, dyl, None, lng, dw0, dbn0, R, D)
    o.scatter_add_rows(lng, idx, ref, D, R, D)
    got = dx.clone()
    o.layernorm_bwd(x, w, mean, rstd, dyl, got, got, dw1, dbn1, R, D, ridx=idx)
    assert rel_err(got, ref) < 1e-6 and rel_err(dw1, dw0) < 1e-5 and rel_err(dbn1, dbn0) < 1e-5
    keep = torch.ones(T, dtype=torch.bool, device=DEV)
    keep[idx] = False
    assert torch.equal(got[keep], dx[keep])


@pytest.mark.parametrize("K,rows_a,rows_b", [(65536, 24, 300), (20480, 8, 40), (4096, 6, 20), (520, 5, 3)])
def test_centering_kernels_on_bf16_logit_rows(K, rows_a, rows_b):
    """lt_softmax_stats_colsum_bf16 / lt_ce_fwd_bwd_logits_bf16 (the `bf16_logits` option: logit rows held in bf16, arithmetic in fp32): fed
    bf16 rows, they have to return what the fp32 entry points return on the SAME (bf16-representable) values -- identical loads after the
    widening, identical arithmetic: statistics, column sums, loss terms and d-logits bit for bit."""
    o = ops()
    g = torch.Generator().manual_seed(K + rows_a)
    Rt = rows_a + rows_b
    tl16 = (torch.randn(Rt, K, generator=g) * 0.3).to(torch.bfloat16).to(DEV)
    tl32 = tl16.float()
    ca = (torch.randn(K, generator=g) * 0.05).to(DEV)
    cb = (torch.randn(K, generator=g) * 0.05).to(DEV)
    itt = 1 / 0.05
    sws = torch.empty(256 * K, device=DEV)
    res = {}
    for name, tl in (("f32", tl32), ("bf16", tl16)):
        stats = torch.zeros(Rt, 2, device=DEV)
        cs_a, cs_b = torch.full((K,), 7.0, device=DEV), torch.full((K,), 7.0, device=DEV)
        o.softmax_stats_colsum(tl[:rows_a], ca, stats[:rows_a], cs_a, rows_a, K, itt, sws)
        o.softmax_stats_colsum(tl[rows_a:], cb, stats[rows_a:], cs_b, rows_b, K, itt, sws)
        R = 24
        gs = torch.Generator().manual_seed(7)
        s16 = torch.randn(R, K, generator=gs).to(torch.bfloat16).to(DEV)
        sx = s16 if name == "bf16" else s16.float()
        ta = torch.randint(0, Rt, (R,), generator=gs, dtype=torch.int32).to(DEV)
        tb = torch.where(torch.rand(R, generator=gs) < 0.5, torch.randint(0, rows_a, (R,), generator=gs, dtype=torch.int32), torch.full((R,), -1, dtype=torch.int32)).to(DEV)
        w = torch.rand(R, generator=gs).to(DEV)
        slot = torch.randint(0, 3, (R,), generator=gs, dtype=torch.int32).to(DEV)
        loss = torch.zeros(5, device=DEV); dl = torch.empty(R, K, device=DEV, dtype=torch.bfloat16)
        o.ce_fwd_bwd_logits(sx, tl, stats, ca, cb, rows_a, ta, tb, w, 0.37, 10.0, itt, loss, dl, R, K, slot=slot)
        res[name] = (stats, cs_a, cs_b, loss, dl)
    torch.cuda.synchronize()
    for a_, b_, what in zip(res["f32"], res["bf16"], ("stats", "colsum a", "colsum b", "loss", "dlogits")):
        if K >= 8192 or what not in ("colsum a", "colsum b"):
            assert torch.equal(a_, b_), what
        else:       # generic widths: the bf16 column sums come from lt_colsum_bf16 (another summation order)
            assert rel_err(b_, a_) < 1e-5, what
    z = torch.cat([(tl32[:rows_a] - ca) * itt, (tl32[rows_a:] - cb) * itt]).double()
    assert torch.allclose(res["bf16"][0][:, 0].double(), z.max(-1).values, atol=1e-5)
    assert rel_err(res["bf16"][1], tl32[:rows_a].double().sum(0)) < 1e-5


@pytest.mark.parametrize("K,rows_a,rows_b", [(65536, 24, 300), (65536, 256, 700), (20484, 8, 40), (4096, 6, 20), (520, 5, 0)])
def test_centering_without_the_probability_matrix(K, rows_a, rows_b):
    """lt_softmax_stats_colsum + lt_ce_fwd_bwd_logits (the softmax-centering path of the step: row statistics and column sums in one pass,
    probabilities rebuilt inside the cross-entropy) against torch and against the three-pass form (lt_softmax_center, lt_colsum_f32,
    lt_ce_fwd_bwd): two centers (DINO cls rows / iBOT patch rows), one- and two-target student rows, slots and row weights."""
    o = ops()
    g = torch.Generator().manual_seed(K + rows_a)
    Rt = rows_a + rows_b
    tl = (torch.randn(Rt, K, generator=g) * 0.3).to(DEV)
    ca = (torch.randn(K, generator=g) * 0.05).to(DEV)
    cb = (torch.randn(K, generator=g) * 0.05).to(DEV)
    itt = 1 / 0.05
    stats = torch.zeros(Rt, 2, device=DEV)
    sws = torch.empty(256 * K if K != 20484 else 3 * K, device=DEV)     # 3 * K: fewer workgroups, longer row walks
    cs_a, cs_b = torch.full((K,), 7.0, device=DEV), torch.full((K,), 7.0, device=DEV)
    o.softmax_stats_colsum(tl[:rows_a], ca, stats[:rows_a], cs_a, rows_a, K, itt, sws)
    o.softmax_stats_colsum(tl[rows_a:], cb, stats[rows_a:], cs_b, rows_b, K, itt, sws)
    z = torch.cat([(tl[:rows_a] - ca) * itt, (tl[rows_a:] - cb) * itt]).double()
    assert torch.allclose(stats[:, 0].double(), z.max(-1).values, atol=1e-5)
    assert rel_err(1.0 / stats[:, 1], torch.exp(z - z.max(-1, keepdim=True).values).sum(-1)) < 1e-5
    assert rel_err(cs_a, tl[:rows_a].double().sum(0)) < 1e-5 and rel_err(cs_b, tl[rows_a:].double().sum(0)) < 2e-5
    ref_a, ref_b = torch.empty(K, device=DEV), torch.empty(K, device=DEV)
    o.colsum_f32(tl[:rows_a], ref_a, rows_a, K)
    assert rel_err(cs_a, ref_a) < 1e-5
    if rows_b:
        o.colsum_f32(tl[rows_a:], ref_b, rows_b, K)
        assert rel_err(cs_b, ref_b) < 1e-5
    else:
        assert float(cs_b.abs().max()) == 0.0       # an empty row range overwrites the sums with zeros
    cs2 = torch.empty(K, device=DEV)
    o.softmax_stats_colsum(tl[:rows_a], ca, torch.zeros(rows_a, 2, device=DEV), cs2, rows_a, K, itt, sws)
    assert torch.equal(cs2, cs_a), "column sums must be bit-reproducible"
    # student rows: one target (cls / patch rows) and two targets (local crops against both global teachers)
    R = 40
    s = torch.randn(R, K, generator=g).to(DEV)
    ta = torch.randint(0, Rt, (R,), generator=g, dtype=torch.int32).to(DEV)
    tb = torch.where(torch.rand(R, generator=g) < 0.5, torch.randint(0, rows_a, (R,), generator=g, dtype=torch.int32), torch.full((R,), -1, dtype=torch.int32)).to(DEV)
    w = torch.rand(R, generator=g).to(DEV)
    slot = torch.randint(0, 3, (R,), generator=g, dtype=torch.int32).to(DEV)
    loss = torch.zeros(5, device=DEV); dl = torch.empty(R, K, device=DEV, dtype=torch.bfloat16)
    o.ce_fwd_bwd_logits(s, tl, stats, ca, cb, rows_a, ta, tb, w, 0.37, 10.0, itt, loss, dl, R, K, slot=slot)
    probs = torch.empty(Rt, K, device=DEV)
    o.softmax_center(tl[:rows_a], ca, probs[:rows_a], rows_a, K, itt)
    if rows_b:
        o.softmax_center(tl[rows_a:], cb, probs[rows_a:], rows_b, K, itt)
    loss3 = torch.zeros(5, device=DEV); dl3 = torch.empty(R, K, device=DEV, dtype=torch.bfloat16)
    o.ce_fwd_bwd(s, probs, ta, tb, w, 0.37, 10.0, loss3, dl3, R, K, slot=slot)
    assert torch.allclose(loss, loss3, rtol=2e-6, atol=1e-6), (loss, loss3)
    assert rel_err(dl, dl3) < 8e-3      # both rounded to bf16 from values that differ in the last fp32 bits
    sr = s.double().requires_grad_(True)
    pr = torch.softmax(z, -1)
    t = pr[ta.long()] + pr[tb.clamp_min(0).long()] * (tb >= 0)[:, None]
    rows_loss = -(0.37 * w.double() * (t * F.log_softmax(sr * 10.0, -1)).sum(-1))
    rows_loss.sum().backward()
    ref_slots = torch.zeros(5, dtype=torch.float64, device=DEV).index_add_(0, slot.long(), rows_loss.detach())
    assert torch.allclose(loss.double(), ref_slots, rtol=2e-5, atol=1e-6)
    assert rel_err(dl, sr.grad) < 8e-3


@pytest.mark.parametrize("K", [4096, 65536, 20484])   # 256-thread kernels / register-resident rows (full and ragged)
def test_losses(K):
    o = ops()
    rows = 24
    g = torch.Generator().manual_seed(5)
    logits = (torch.randn(rows, K, generator=g) * 0.3).to(DEV)
    center = (torch.randn(K, generator=g) * 0.05).to(DEV)
    probs = torch.empty(rows, K, device=DEV)
    o.softmax_center(logits, center, probs, rows, K, 1 / 0.04)
    ref = F.softmax((logits - center) / 0.04, -1)
    assert rel_err(probs, ref) < 1e-5
    cs = torch.empty(K, device=DEV)
    o.colsum_f32(logits, cs, rows, K)
    c2 = center.clone()
    o.center_ema(c2, cs, 1.0 / rows, 0.9, K)
    assert torch.allclose(c2, center * 0.9 + logits.mean(0) * 0.1, atol=1e-6)
    # CE with two teachers and row weights
    s = (torch.randn(16, K, generator=g)).to(DEV).requires_grad_(True)
    ta = torch.arange(16, dtype=torch.int32, device=DEV) % 12
    tb = (torch.arange(16, dtype=torch.int32, device=DEV) % 12) + 12
    w = torch.rand(16, generator=g).to(DEV)
    loss = torch.zeros(1, device=DEV); dl = torch.empty(16, K, device=DEV, dtype=torch.bfloat16)
    o.ce_fwd_bwd(s.detach(), probs, ta, tb, w, 0.37, 10.0, loss, dl, 16, K)
    t = probs[ta.long()] + probs[tb.long()]
    ref_loss = -(0.37 * w * (t * F.log_softmax(s * 10.0, -1)).sum(-1)).sum()
    ref_loss.backward()
    assert abs(loss.item() - ref_loss.item()) < 1e-4 * abs(ref_loss.item())
    assert rel_err(dl, s.grad) < 8e-3
    # Sinkhorn-Knopp (single rank): same recurrence as dinov2_loss.py:84-115
    Q = torch.empty(rows, K, device=DEV)
    o.sk_exp(logits, Q, 1 / 0.05)
    colsum = torch.empty(K, device=DEV)
    for it in range(3):
        o.colsum_f32(Q, colsum, rows, K)
        o.sk_iter(Q, colsum, rows, K, float(rows), float(rows) if it == 2 else 1.0)
    q = torch.exp(logits / 0.05).t(); q /= q.sum()
    for _ in range(3):
        q /= q.sum(1, keepdim=True); q /= K; q /= q.sum(0, keepdim=True); q /= rows
    q *= rows
    assert rel_err(Q, q.t()) < 1e-4


def test_koleo():
    o = ops()
    n, D = 48, 96
    g = torch.Generator().manual_seed(6)
    x = torch.randn(n + 3, D, generator=g).to(DEV)[:n]
    loss = torch.zeros(1, device=DEV); dx = torch.zeros(n, D, device=DEV)
    ws = torch.empty(2 * n * D + 2 * n, device=DEV); nn = torch.empty(n, dtype=torch.int32, device=DEV)
    o.koleo_fwd_bwd(x, D, loss, dx, D, n, D, 0.1, ws, nn)
    xr = x.clone().requires_grad_(True)
    xn = F.normalize(xr, p=2, dim=-1, eps=1e-8)
    cos = (xn @ xn.t()).clone(); cos.fill_diagonal_(-2)
    idx = cos.argmax(1)
    ref = -torch.log(torch.linalg.vector_norm(xn - xn[idx] + 1e-8, dim=-1) + 1e-8).mean() * 0.1
    ref.backward()
    assert torch.equal(nn.long(), idx)
    assert abs(loss.item() - ref.item()) < 1e-5
    assert rel_err(dx, xr.grad) < 1e-4


def test_koleo_hand_computed_value():
    """KoLeo is un-vendored LightlySSL code (parity unpinned): four points done on paper.  Rows (3,4), (4,3), (0,5), (5,0) normalise
    to (.6,.8), (.8,.6), (0,1), (1,0); nearest neighbours by cosine: 0<->1 (cos .96), 2->0 (.8), 3->1 (.8); distances sqrt(.08), sqrt(.08),
    sqrt(.4), sqrt(.4); loss = -(ln .08 + ln .4) / 4 = -ln(.032) / 4 = 0.8605048...  The gradient is tangent to the unit circle, so
    dL/dx . x = 0 for every row (scale invariance of the normalisation)."""
    o = ops()
    n, D = 4, 64
    x = torch.zeros(n, D, device=DEV)
    x[:, :2] = torch.tensor([[3.0, 4.0], [4.0, 3.0], [0.0, 5.0], [5.0, 0.0]], device=DEV)
    loss = torch.zeros(1, device=DEV); dx = torch.zeros(n, D, device=DEV)
    ws = torch.empty(2 * n * D + 2 * n, device=DEV); nn = torch.empty(n, dtype=torch.int32, device=DEV)
    o.koleo_fwd_bwd(x, D, loss, dx, D, n, D, 1.0, ws, nn)
    assert nn.tolist() == [1, 0, 0, 1]
    assert loss.item() == pytest.approx(-math.log(0.032) / 4, rel=1e-5)
    assert float((dx * x).sum(1).abs().max()) < 1e-5
    assert float(dx[:, 2:].abs().max()) < 1e-8      # (PairwiseDistance adds its eps to every coordinate difference: ~1e-9 in the unused dimensions)
    # row 2 = (0, 5): L_2 = -ln|z - x0| / 4 with z = (0,1), x0 = (.6,.8); dL/dz = -(z - x0) / (4 |z - x0|^2) = -(-.6, .2) / 1.6; projected on the tangent
    # (1, 0) and divided by |row| = 5: dL/dx_2 = (.6 / 1.6 / 5, 0) = (0.075, 0) -- plus row 2's role as nobody's neighbour: none.
    assert dx[2, 0].item() == pytest.approx(0.075, rel=1e-4) and abs(dx[2, 1].item()) < 1e-6


def test_adamw_ema_match_torch():
    o = ops()
    sizes = [1024 * 3, 1024, 2048]
    n = sum(sizes)
    g = torch.Generator().manual_seed(7)
    p = torch.randn(n, generator=g).to(DEV); p0 = p.clone()
    seg_of_chunk = torch.tensor([0, 0, 0, 1, 2, 2], dtype=torch.int32, device=DEV)
    seg_lr = torch.tensor([1e-3, 5e-4, 2e-3], device=DEV)
    seg_wd = torch.tensor([1, 0, 1], dtype=torch.uint8, device=DEV)
    seg_fr = torch.tensor([0, 0, 1], dtype=torch.uint8, device=DEV)
    m = torch.zeros(n, device=DEV); v = torch.zeros(n, device=DEV); pb = torch.empty(n, device=DEV, dtype=torch.bfloat16)
    views = list(p0.clone().split(sizes))
    params = [torch.nn.Parameter(t) for t in views]
    opt = torch.optim.AdamW([{"params": [params[0]], "lr": 1e-3, "weight_decay": 0.04},
                             {"params": [params[1]], "lr": 5e-4, "weight_decay": 0.0},
                             {"params": [params[2]], "lr": 2e-3, "weight_decay": 0.04}], betas=(0.9, 0.999), eps=1e-8)
    for step in range(1, 4):
        grad = torch.randn(n, generator=g).to(DEV) * 3
        ss = torch.zeros(1, device=DEV)
        o.sumsq(grad, ss)
        assert abs(ss.item() - (grad ** 2).sum().item()) < 1e-4 * ss.item()
        freeze = step == 1
        o.adamw_flat(p, grad, m, v, pb, seg_of_chunk, seg_lr, seg_wd, seg_fr, freeze, 0.5, 0.04, 0.9, 0.999, 1e-8, step, ss, 3.0)
        for prm, gpart in zip(params, grad.split(sizes)):
            prm.grad = gpart.clone()
        torch.nn.utils.clip_grad_norm_(params, 3.0)
        for gi, grp in enumerate(opt.param_groups):
            grp["lr"] = [1e-3, 5e-4, 2e-3][gi] * 0.5
            if gi == 2 and freeze:
                grp["lr"] = 0.0
        opt.step()
        ref = torch.cat([q.detach() for q in params])
        assert (p - ref).abs().max().item() < 2e-6, f"adamw step {step}"
        assert torch.equal(pb.float(), ref.to(torch.bfloat16).float()) or (pb.float() - ref).abs().max().item() < 1e-2
    t = torch.randn(n, generator=g).to(DEV); t0 = t.clone(); tb = torch.empty(n, device=DEV, dtype=torch.bfloat16)
    o.ema_flat(t, p, tb, 0.992)
    assert torch.allclose(t, t0 * 0.992 + p * (1 - 0.992), atol=1e-6)


def test_sumsq_is_deterministic_and_accumulates():
    """The global gradient norm feeds the clip coefficient of every parameter: identical gradients must give identical bits
    (data-parallel replicas would otherwise drift apart), on any stream, and `out` accumulates across calls."""
    o = ops()
    g = torch.Generator().manual_seed(11)
    for n in (1000, 1024 * 1024 + 4, 37 * 1024 * 1024):   # one block, ragged tail, more chunks than the 1024-block grid
        x = (torch.randn(n, generator=g) * 3).to(DEV)
        outs = []
        for rep in range(6):
            ss = torch.zeros(1, device=DEV)
            torch.cuda.synchronize()
            if rep % 2:
                with torch.cuda.stream(torch.cuda.Stream()):
                    o.sumsq(x, ss)
            else:
                o.sumsq(x, ss)
            torch.cuda.synchronize()
            outs.append(ss.clone())
        assert all(torch.equal(outs[0], t) for t in outs[1:]), [t.item() for t in outs]
        ref = (x.double() ** 2).sum().item()
        assert abs(outs[0].item() - ref) < 2e-6 * ref
        ss = torch.full((1,), 5.0, device=DEV)
        o.sumsq(x, ss)
        assert abs(ss.item() - 5.0 - ref) < 2e-6 * ref + 1e-3


# ------------------------------------------------------------------------------------------ convolutional student ops (csrc/conv.hip)
def _nhwc(x):   # [B,C,H,W] -> NHWC rows [B*H*W, C]
    return x.permute(0, 2, 3, 1).reshape(-1, x.shape[1]).contiguous()


@pytest.mark.parametrize("B,C,H,W,k,s,p", [(2, 16, 9, 11, 3, 1, 1), (3, 8, 12, 12, 3, 2, 1), (2, 32, 7, 7, 1, 2, 0), (1, 24, 5, 6, 3, 2, 1)])
def test_im2col_col2im_nhwc(B, C, H, W, k, s, p):
    """im2col (taps outer, channels inner) and its transpose against F.unfold / F.fold (which order columns channel-major)."""
    o = ops()
    g = torch.Generator().manual_seed(B * C + H)
    x = bf(torch.randn(B, C, H, W, generator=g))
    Ho, Wo = o.conv_out_size(H, k, s, p), o.conv_out_size(W, k, s, p)
    cols = torch.empty(B * Ho * Wo, k * k * C, device=DEV, dtype=torch.bfloat16)
    o.im2col_nhwc(_nhwc(x).to(DEV), cols, B, H, W, C, k, k, s, p)
    ref = F.unfold(x.float(), k, padding=p, stride=s)                     # [B, C*k*k, L] (c outer, taps inner)
    ref = ref.view(B, C, k * k, Ho * Wo).permute(0, 3, 2, 1).reshape(B * Ho * Wo, k * k * C)
    assert torch.equal(cols.float().cpu(), ref)
    d = bf(torch.randn(B * Ho * Wo, k * k * C, generator=g))
    add = bf(torch.randn(B * H * W, C, generator=g))
    dx = torch.empty(B * H * W, C, device=DEV, dtype=torch.bfloat16)
    o.col2im_nhwc(d.to(DEV), dx, B, H, W, C, k, k, s, p, add=add.to(DEV))
    dref = d.float().view(B, Ho * Wo, k * k, C).permute(0, 3, 2, 1).reshape(B, C * k * k, Ho * Wo)
    fold = F.fold(dref, (H, W), k, padding=p, stride=s)
    want = _nhwc(fold) + add.float()
    assert rel_err(dx, want) < 1e-2


def test_im2col_nchw_stem():
    o = ops()
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 3, 20, 22, generator=g)
    Ho, Wo = o.conv_out_size(20, 7, 2, 3), o.conv_out_size(22, 7, 2, 3)
    cols = torch.empty(2 * Ho * Wo, 152, device=DEV, dtype=torch.bfloat16)
    o.im2col_nchw_f32(x.to(DEV), cols, 7, 7, 2, 3)
    ref = F.unfold(x, 7, padding=3, stride=2).permute(0, 2, 1).reshape(2 * Ho * Wo, 147)     # c outer, (ky,kx) inner = weight.flatten(1)
    assert torch.equal(cols[:, :147].float().cpu(), bf(ref).float()) and cols[:, 147:].abs().max().item() == 0


@pytest.mark.parametrize("rows,C,relu,resid", [(1000, 64, True, False), (77, 8, True, True), (4096, 256, False, False), (300, 2048, True, True),
                                               (50000, 64, True, False)])
def test_batchnorm_fwd_bwd(rows, C, relu, resid):
    """Training-mode BatchNorm over rows (+ ReLU, + residual add) against F.batch_norm in fp32, running statistics included."""
    o = ops()
    g = torch.Generator().manual_seed(rows + C)
    x = bf(torch.randn(rows, C, generator=g) * 1.5 + 0.3)
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.2
    r = bf(torch.randn(rows, C, generator=g)) if resid else None
    rm, rv = torch.zeros(C), torch.ones(C)
    xr = x.float().clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rr = r.float().clone().requires_grad_(True) if resid else None
    pre = F.batch_norm(xr, rm, rv, gr, br, training=True, momentum=0.1, eps=1e-5)
    if resid:
        pre = pre + rr
    ref = F.relu(pre) if relu else pre
    y = torch.empty(rows, C, device=DEV, dtype=torch.bfloat16)
    mean, rstd = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    ws = torch.empty(o.batchnorm_ws_floats(C), device=DEV)
    rmd, rvd = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    o.batchnorm_fwd(x.to(DEV), gamma.to(DEV), beta.to(DEV), y, mean, rstd, rows, C, ws, resid=r.to(DEV) if resid else None, running_mean=rmd,
                    running_var=rvd, relu=relu)
    assert rel_err(y, ref) < 1e-2
    assert rel_err(rmd, rm) < 1e-4 and rel_err(rvd, rv) < 1e-4
    assert rel_err(mean, x.float().mean(0)) < 1e-4
    dy = bf(torch.randn(rows, C, generator=g))
    ref.backward(dy.float())
    dz = torch.empty(rows, C, device=DEV, dtype=torch.bfloat16) if relu else None
    dx = torch.empty(rows, C, device=DEV, dtype=torch.bfloat16)
    dg, db = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    o.batchnorm_bwd(dy.to(DEV), x.to(DEV), gamma.to(DEV), mean, rstd, dx, rows, C, ws, y=y if relu else None, dz=dz, dgamma=dg, dbeta=db)
    # the ReLU mask comes from the bf16-rounded output: elements whose pre-activation rounds to +-0 may differ -> norm-wise check
    assert rel_err(dg, gr.grad) < 2e-2 and rel_err(db, br.grad) < 2e-2
    assert rel_err(dx, xr.grad) < 3e-2
    if resid and relu:
        assert rel_err(dz, rr.grad) < 1e-2      # dz is also the gradient of the residual input


@pytest.mark.parametrize("B,C,H,W", [(2, 16, 12, 12), (1, 8, 7, 9), (3, 64, 16, 16)])
def test_maxpool3x3s2(B, C, H, W):
    o = ops()
    g = torch.Generator().manual_seed(C + H)
    x = bf(torch.relu(torch.randn(B, C, H, W, generator=g)))           # ReLU output: ties at 0 exercise "first maximum wins"
    xr = x.float().clone().requires_grad_(True)
    ref = F.max_pool2d(xr, 3, 2, 1)
    Ho, Wo = ref.shape[2], ref.shape[3]
    y = torch.empty(B * Ho * Wo, C, device=DEV, dtype=torch.bfloat16)
    idx = torch.empty(B * Ho * Wo, C, device=DEV, dtype=torch.uint8)
    o.maxpool3x3s2_fwd(_nhwc(x).to(DEV), y, idx, B, H, W, C)
    assert torch.equal(y.float().cpu(), _nhwc(ref.detach()).float())
    dy = bf(torch.randn(B, C, Ho, Wo, generator=g))
    ref.backward(dy.float())
    dx = torch.empty(B * H * W, C, device=DEV, dtype=torch.bfloat16)
    o.maxpool3x3s2_bwd(_nhwc(dy).to(DEV), idx, dx, B, H, W, C)
    assert rel_err(dx, _nhwc(xr.grad)) < 1e-2


def test_token_mean_and_pool_backward():
    o = ops()
    g = torch.Generator().manual_seed(1)
    B, n, C = 5, 49, 64
    x = bf(torch.randn(B, n, C, generator=g))
    out = torch.empty(B, C, device=DEV, dtype=torch.bfloat16)
    o.token_mean(x.reshape(B * n, C).to(DEV), out, B, n, C)
    assert rel_err(out, x.float().mean(1)) < 1e-2
    dt, dp = torch.randn(B * n, C, generator=g), torch.randn(B, C, generator=g)
    d = torch.empty(B * n, C, device=DEV, dtype=torch.bfloat16)
    o.pool_bwd_add(dt.to(DEV), dp.to(DEV), d, B, n, C)
    assert rel_err(d, (dt.view(B, n, C) + dp[:, None] / n).reshape(B * n, C)) < 1e-2
    a, b_ = bf(torch.randn(64, 8, generator=g)), bf(torch.randn(64, 8, generator=g))
    s = torch.empty(64, 8, device=DEV, dtype=torch.bfloat16)
    o.add_bf16(a.to(DEV), b_.to(DEV), s)
    assert rel_err(s, a.float() + b_.float()) < 1e-2


@pytest.mark.gpu
@pytest.mark.parametrize("n", [8, 4096, 24 * 2048 + 8])
def test_gelu_elementwise_fwd_bwd(n):
    """lt_gelu_fwd_bf16 / lt_gelu_bwd_bf16 (the activation of the BatchNorm projection heads) against torch's erf GELU in fp32."""
    from lightly_train_amd import ops
    g = torch.Generator().manual_seed(n)
    x = (torch.randn(n, generator=g) * 2).to(torch.bfloat16).cuda()
    dy = torch.randn(n, generator=g).to(torch.bfloat16).cuda()
    y = torch.empty_like(x); dx = torch.empty_like(x)
    ops.gelu_fwd(x, y, n)
    ops.gelu_bwd(dy, x, dx, n)
    xf = x.float().requires_grad_(True)
    ref = torch.nn.functional.gelu(xf)
    ref.backward(dy.float())
    assert torch.allclose(y.float(), ref.detach(), atol=2e-2, rtol=1e-2)
    assert torch.allclose(dx.float(), xf.grad, atol=2e-2, rtol=1e-2)
    ops.gelu_bwd(dy, x, dy, n)      # in place
    assert torch.equal(dy, dx)


@pytest.mark.parametrize("nesterov,dampening,clip", [(False, 0.0, 0.0), (True, 0.0, 1.0), (False, 0.3, 0.5)])
def test_lars_flat_matches_oracle(nesterov, dampening, clip):
    """lt_lars_norms + lt_lars_flat on flat storage against oracle/lars_oracle.py (fp32 CPU), 3 steps: weight-decay and no-decay groups, a
    tensor that is all zero (no trust ratio: plain step), a tensor whose gradient is zero, gradient clipping folded into the step."""
    from lightly_train_amd import ops
    from lightly_train_amd.lars import FlatLARS, LARSArgs
    from lightly_train_amd.params import FlatParams
    from oracle.lars_oracle import LARS

    g = torch.Generator().manual_seed(5)
    shapes = [("w1", (64, 48)), ("b1", (64,)), ("w2", (700, 3)), ("zero", (33, 9)), ("nograd", (16, 16)), ("w3", (2100,))]
    named = [(n, torch.zeros(sh) if n == "zero" else torch.randn(sh, generator=g) * 0.1) for n, sh in shapes]
    decay = {"w1": True, "b1": False, "w2": True, "zero": True, "nograd": True, "w3": False}
    fp = FlatParams(named, "cuda", True)
    args = LARSArgs(lr=0.7, momentum=0.9, dampening=dampening, weight_decay=1e-3, nesterov=nesterov, trust_coefficient=0.01)
    opt = FlatLARS(fp, args)
    seg_lr = torch.full((len(named),), args.lr, device="cuda")
    seg_wd = torch.tensor([1 if decay[n] else 0 for n, _ in named], dtype=torch.uint8, device="cuda")
    ref_p = {n: t.clone().requires_grad_(True) for n, t in named}
    ref = LARS([{"params": [ref_p[n] for n, _ in named if decay[n]]}, {"params": [ref_p[n] for n, _ in named if not decay[n]], "weight_decay": 0.0}],
               lr=args.lr, momentum=args.momentum, dampening=dampening, weight_decay=args.weight_decay, nesterov=nesterov,
               trust_coefficient=args.trust_coefficient, eps=args.eps)
    sumsq = torch.zeros(1, device="cuda")
    for step in range(3):
        for n, t in named:
            gr = torch.zeros_like(t) if n == "nograd" else torch.randn(t.shape, generator=g)
            fp.g[n].copy_(gr)
            ref_p[n].grad = gr.clone()
        factor = 0.5 + 0.25 * step
        if clip > 0:
            torch.nn.utils.clip_grad_norm_(list(ref_p.values()), clip)
        for gr_ in ref.param_groups:
            gr_["lr"] = args.lr * factor
        ref.step()
        sumsq.zero_()
        ops.sumsq(fp.grad, sumsq)
        opt.step(seg_lr, seg_wd, factor, sumsq if clip > 0 else None, clip)
        for n, _ in named:
            assert torch.allclose(fp.p[n].cpu(), ref_p[n].detach(), rtol=2e-5, atol=2e-6), (step, n)
            assert torch.equal(fp.b[n].float().cpu(), fp.p[n].to(torch.bfloat16).float().cpu()), n
    assert torch.equal(fp.p["nograd"].cpu(), named[4][1])     # zero gradient: no trust ratio, no decay, no step


@pytest.mark.parametrize("rows,C,relu", [(96, 64, True), (4100, 256, False)])
def test_sync_batchnorm_halves_equal_the_whole(rows, C, relu):
    """SyncBatchNorm (lt_batchnorm_stats / _fwd_from_sums / _bwd_sums / _bwd_from_sums around the caller's all-reduce): two "ranks" holding
    the two parts of a batch -- the all-reduce played by adding the other part's sums -- give the outputs, running estimates, input
    gradients and (summed) parameter gradients of training-mode BatchNorm over the whole batch."""
    from lightly_train_amd import ops
    g = torch.Generator().manual_seed(rows)
    x = (torch.randn(rows, C, generator=g) * 1.5 + 0.3).to(torch.bfloat16).cuda()
    dy = torch.randn(rows, C, generator=g).to(torch.bfloat16).cuda()
    gamma = (1 + 0.2 * torch.randn(C, generator=g)).cuda(); beta = (0.1 * torch.randn(C, generator=g)).cuda()
    ws = torch.empty(ops.batchnorm_ws_floats(C), device="cuda")
    cut = rows // 3 // 8 * 8 + 8          # unequal parts
    parts = [(0, cut), (cut, rows - cut)]

    def whole():
        y = torch.empty_like(x); dx = torch.empty_like(x); dz = torch.empty_like(x)
        mean = torch.empty(C, device="cuda"); rstd = torch.empty(C, device="cuda")
        rm = torch.zeros(C, device="cuda"); rv = torch.ones(C, device="cuda")
        dg = torch.zeros(C, device="cuda"); db = torch.zeros(C, device="cuda")
        ops.batchnorm_fwd(x, gamma, beta, y, mean, rstd, rows, C, ws, running_mean=rm, running_var=rv, relu=relu)
        ops.batchnorm_bwd(dy, x, gamma, mean, rstd, dx, rows, C, ws, y=y if relu else None, dz=dz if relu else None, dgamma=dg, dbeta=db)
        return y, dx, rm, rv, dg, db

    def split():
        y = torch.empty_like(x); dx = torch.empty_like(x); dz = torch.empty_like(x)
        rms = [torch.zeros(C, device="cuda") for _ in parts]; rvs = [torch.ones(C, device="cuda") for _ in parts]
        means = [torch.empty(C, device="cuda") for _ in parts]; rstds = [torch.empty(C, device="cuda") for _ in parts]
        dg = torch.zeros(C, device="cuda"); db = torch.zeros(C, device="cuda")
        for phase in ("fwd", "bwd"):
            own = []
            for (r0, n) in parts:      # first pass: every part's own sums (what it would contribute to the all-reduce)
                cap = []
                def grab(t, cap=cap):
                    cap.append(t.clone()); t.fill_(float("nan"))      # the result of this pass is discarded
                    t[-1] = n; t[:-1] = 0; t[C:2 * C] = 1
                if phase == "fwd":
                    ops.batchnorm_fwd(x[r0:r0 + n], gamma, beta, torch.empty_like(x[r0:r0 + n]), torch.empty(C, device="cuda"), torch.empty(C, device="cuda"), n, C, ws, sync=grab)
                else:
                    i = parts.index((r0, n))
                    ops.batchnorm_bwd(dy[r0:r0 + n], x[r0:r0 + n], gamma, means[i], rstds[i], torch.empty_like(x[r0:r0 + n]), n, C, ws,
                                      y=y[r0:r0 + n] if relu else None, dz=torch.empty_like(x[r0:r0 + n]) if relu else None, sync=grab)
                own.append(cap[0])
            total = own[0] + own[1]
            for i, (r0, n) in enumerate(parts):
                allreduce = lambda t: t.copy_(total)
                if phase == "fwd":
                    ops.batchnorm_fwd(x[r0:r0 + n], gamma, beta, y[r0:r0 + n], means[i], rstds[i], n, C, ws, running_mean=rms[i], running_var=rvs[i],
                                      relu=relu, sync=allreduce)
                else:
                    ops.batchnorm_bwd(dy[r0:r0 + n], x[r0:r0 + n], gamma, means[i], rstds[i], dx[r0:r0 + n], n, C, ws, y=y[r0:r0 + n] if relu else None,
                                      dz=dz[r0:r0 + n] if relu else None, dgamma=dg, dbeta=db, sync=allreduce)
        assert torch.equal(rms[0], rms[1]) and torch.equal(rvs[0], rvs[1])       # every rank ends with the same running estimates
        return y, dx, rms[0], rvs[0], dg, db

    yw, dxw, rmw, rvw, dgw, dbw = whole()
    ys, dxs, rm_s, rv_s, dgs, dbs = split()
    assert (ys.float() - yw.float()).abs().max().item() <= 2e-2 and float((ys.float() - yw.float()).abs().mean()) < 1e-4    # rounding flips only
    assert float((dxs.float() - dxw.float()).abs().mean()) < 1e-4 and (dxs.float() - dxw.float()).abs().max().item() <= 3e-2
    assert torch.allclose(rm_s, rmw, atol=1e-6) and torch.allclose(rv_s, rvw, rtol=1e-5, atol=1e-6)
    assert torch.allclose(dgs, dgw, rtol=2e-3, atol=2e-2) and torch.allclose(dbs, dbw, rtol=2e-3, atol=2e-2)


def test_comm_handle_single_rank_allreduce_is_ordered_with_the_streams():
    """lt_comm_* (the C ABI's RCCL communicator handle) with one rank: the all-reduce is the identity in value; what is checked is the
    fencing -- it runs on the communicator's stream AFTER the producer enqueued on the caller's stream, and a consumer that called
    lt_comm_wait sees its result."""
    from lightly_train_amd.parallel import AbiComm

    comm = AbiComm(0, 1, AbiComm.unique_id())
    try:
        assert comm.lib.lt_comm_size() == 1
        n = 64 * 1024 * 1024
        buf = torch.zeros(n, device=DEV)
        side = torch.cuda.Stream()
        for it in range(3):
            with torch.cuda.stream(side):
                buf.fill_(float(it + 1))          # producer on `side`
                comm.all_reduce(buf)              # ordered after it, on the communicator's stream
            comm.wait()                           # the current (default) stream waits for the collective
            out = buf * 2.0                       # consumer
            assert float(out[0]) == 2.0 * (it + 1) and float(out[-1]) == 2.0 * (it + 1)
            side.wait_stream(torch.cuda.current_stream())
    finally:
        comm.destroy()
    assert comm.lib.lt_comm_size() == 0


@pytest.mark.parametrize("M,N,K", [(50432, 768, 768), (2500, 768, 3072), (2048, 384, 384), (4099, 1024, 1024), (300, 768, 768), (2304, 768, 200)])
def test_residual_gemm_with_the_next_layernorm_behind_it(M, N, K):
    """lt_gemm_desc.ln_* (round 6): LT_EPI_RESID with the next LayerNorm handed to the GEMM call -- the library issues lt_layernorm_fwd behind
    the GEMM, whichever kernel took it (256-row static-address kernel, 128-row kernel at M = 300, partial K-tile at K = 200).  Must equal
    lt_gemm_bf16 followed by lt_layernorm_fwd BIT FOR BIT: the fp32 residual stream, the bf16 operand, mean and rstd; ragged row counts,
    N = 384, 768, 1024.  (LT_GEMM_ROWLN is no longer read: the row-owning kernel mode it switched was removed, profiles/r06_rowln_probe.md.)"""
    o = ops()
    g = torch.Generator().manual_seed(M + N + K)
    A = bf(torch.randn(M, K, generator=g) * 0.5).to(DEV)
    W = bf(torch.randn(N, K, generator=g) * 0.05).to(DEV)
    bias = (torch.randn(N, generator=g) * 0.1).to(DEV)
    gamma = (torch.rand(N, generator=g) + 0.5).to(DEV)
    resid = torch.randn(M, N, generator=g).to(DEV)
    lw, lb = (torch.rand(N, generator=g) + 0.5).to(DEV), (torch.randn(N, generator=g) * 0.1).to(DEV)
    ref_x = torch.empty(M, N, device=DEV)
    o.gemm(A, W, ref_x, M=M, N=N, K=K, epilogue=o.EPI_RESID, bias=bias, gamma=gamma, resid=resid)
    ref_y = torch.empty(M, N, device=DEV, dtype=torch.bfloat16); ref_m = torch.empty(M, device=DEV); ref_r = torch.empty(M, device=DEV)
    o.layernorm_fwd(ref_x, lw, lb, M, N, y_bf16=ref_y, mean=ref_m, rstd=ref_r, eps=1e-6)
    for env in (None, "0"):
        if env is None:
            os.environ.pop("LT_GEMM_ROWLN", None)
        else:
            os.environ["LT_GEMM_ROWLN"] = env
        try:
            x = torch.full((M, N), float("nan"), device=DEV)
            y = torch.zeros(M, N, device=DEV, dtype=torch.bfloat16); mu = torch.zeros(M, device=DEV); rs = torch.zeros(M, device=DEV)
            o.gemm(A, W, x, M=M, N=N, K=K, epilogue=o.EPI_RESID, bias=bias, gamma=gamma, resid=resid, ln=dict(weight=lw, bias=lb, out=y, mean=mu, rstd=rs, eps=1e-6))
            torch.cuda.synchronize()
        finally:
            os.environ.pop("LT_GEMM_ROWLN", None)
        assert torch.equal(x, ref_x), env
        assert torch.equal(y, ref_y) and torch.equal(mu, ref_m) and torch.equal(rs, ref_r), env
    ref = resid + gamma * (A.float() @ W.float().t() + bias)
    assert rel_err(ref_x, ref) < 1e-5
    assert rel_err(ref_y, F.layer_norm(ref, (N,), lw, lb, 1e-6)) < 8e-3


@pytest.mark.parametrize("M,N,K,tb,epi", [(50432, 768, 768, False, "resid"), (50432, 768, 3072, True, "bf16"), (25216, 768, 768, False, "bf16"), (50000, 768, 768, True, "f32"),
                                          (50432, 3072, 768, False, "gelu"), (50432, 3072, 768, True, "gelugrad"), (51200, 768, 2304, True, "bf16"), (66000, 768, 192, False, "resid")])
def test_last_round_as_128_row_tiles_is_bit_identical(M, N, K, tb, epi):
    """gemm128e_kernel (round 6, opt-in: LT_GEMM_TAIL128=1 -- faster alone on the chip, 1 ms slower inside the five-stream step): when the
    256 x 256 tiles of a forward / dgrad GEMM leave a partly filled last round that fits one round of 128 x 256 tiles, the dispatcher gives
    the full rounds to the 256-row kernel and the remaining rows to the 128-row tail kernel.  Same per-accumulator MFMA order: every output
    must equal the single 256-row launch (the default) BIT FOR BIT -- all five epilogues,
    both B layouts, ragged row counts (50 000: the last tile row partial; 66 000: 258 tile rows)."""
    o = ops()
    g = torch.Generator().manual_seed(M + N + K)
    A = bf(torch.randn(M, K, generator=g) * 0.5).to(DEV)
    W = bf(torch.randn(N, K, generator=g) * 0.05).to(DEV)
    b_in = W.t().contiguous() if tb else W
    bias = (torch.randn(N, generator=g) * 0.1).to(DEV)
    kw = {}
    f32 = epi in ("resid", "f32")
    if epi == "resid":
        kw = dict(epilogue=o.EPI_RESID, bias=bias, gamma=(torch.rand(N, generator=g) + 0.5).to(DEV), resid=torch.randn(M, N, generator=g).to(DEV),
                  rowscale=torch.rand(M, generator=g).to(DEV))
    elif epi == "f32":
        kw = dict(epilogue=o.EPI_F32, bias=bias)
    elif epi == "bf16":
        kw = dict(epilogue=o.EPI_BF16, bias=bias)
    elif epi == "gelu":
        kw = dict(epilogue=o.EPI_BF16_GELU, bias=bias)
    else:
        kw = dict(epilogue=o.EPI_BF16_GELUGRAD, aux=bf(torch.randn(M, N, generator=g)).to(DEV))
    outs = []
    for env in ("0", "1"):
        os.environ["LT_GEMM_TAIL128"] = env
        try:
            out = torch.full((M, N), float("nan"), device=DEV, dtype=torch.float32 if f32 else torch.bfloat16)
            out2 = torch.full((M, N), float("nan"), device=DEV, dtype=torch.bfloat16) if epi in ("gelu", "resid") else None
            o.gemm(A, b_in, out, M=M, N=N, K=K, trans_b=tb, out2=out2, **kw)
            torch.cuda.synchronize()
        finally:
            os.environ.pop("LT_GEMM_TAIL128", None)
        outs.append((out, out2))
    assert not torch.isnan(outs[1][0].float()).any()
    assert torch.equal(outs[0][0], outs[1][0])
    if outs[0][1] is not None:
        assert torch.equal(outs[0][1], outs[1][1])
    ref = A[:512].float() @ W.float().t()
    if epi == "f32":
        assert rel_err(outs[1][0][:512], ref + bias) < 1e-5
