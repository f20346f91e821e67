"""TEST INFRASTRUCTURE ONLY: plain-torch CPU stand-ins for a few HIP ops, used to check the ORCHESTRATION of the convolutional
student engine (lightly_train_amd/resnet.py: buffer shapes, layouts, the order of the backward chain) on a box without a GPU.
The kernels themselves are checked against torch on the MI355X (tests/test_gpu_ops.py).  Never imported by the product."""
import contextlib

import torch
import torch.nn.functional as F


def _rows_to_nchw(x, B, H, W, C):
    return x[: B * H * W].float().view(B, H, W, C).permute(0, 3, 1, 2)


def _nchw_to_rows(x):
    return x.permute(0, 2, 3, 1).reshape(-1, x.shape[1])


@contextlib.contextmanager
def emulate(ops):
    saved = {}

    def patch(name, fn):
        saved[name] = getattr(ops, name)
        setattr(ops, name, fn)

    def gemm(a, b, out, *, M, N, K, trans_a=False, trans_b=False, epilogue=0, bias=None, lda=None, ldb=None, ldc=None, split_k=1, workspace=None, **kw):
        A = (a.float().reshape(-1, lda or (M if trans_a else K))[:K, :M].t() if trans_a else a.float().reshape(-1, lda or K)[:M, :K])
        Bm = (b.float().reshape(-1, ldb or (N if trans_b else K))[:K, :N] if trans_b else b.float().reshape(-1, ldb or K)[:N, :K].t())
        r = A @ Bm
        if bias is not None:
            r = r + bias
        o2 = out.reshape(-1, ldc or N)
        if epilogue == ops.EPI_F32_ACCUM:
            o2[:M, :N] += r
        elif epilogue == ops.EPI_BF16_GELU:            # activation + (optionally) the saved pre-activation
            if kw.get("out2") is not None:
                kw["out2"].reshape(-1, N)[:M] = r.to(kw["out2"].dtype)
            o2[:M, :N] = F.gelu(r).to(out.dtype)
        elif epilogue == ops.EPI_BF16_GELUGRAD:        # dX * GELU'(saved pre-activation)
            pre = kw["aux"].reshape(-1, N)[:M].float().detach().requires_grad_(True)
            gp, = torch.autograd.grad(F.gelu(pre).sum(), pre)
            o2[:M, :N] = (r * gp).to(out.dtype)
        else:
            assert epilogue in (ops.EPI_BF16, ops.EPI_F32), epilogue
            o2[:M, :N] = r.to(out.dtype)
        return out

    def l2norm_fwd(x, y, inv, rows, D, eps=1e-12):
        n = x[:rows].float().norm(dim=1).clamp_min(eps)
        inv[:rows] = 1.0 / n
        y[:rows] = (x[:rows].float() / n[:, None]).to(y.dtype)

    def l2norm_bwd(dy, x, inv, dx, rows, D):
        yv = x[:rows].float() * inv[:rows, None]
        d = dy[:rows].float()
        dx[:rows] = ((d - yv * (yv * d).sum(1, keepdim=True)) * inv[:rows, None]).to(dx.dtype)

    def weightnorm_fwd(v, g, w, K, D):
        w.copy_((v * (g / v.norm(dim=1, keepdim=True))).to(w.dtype))

    def weightnorm_bwd(dw, v, g, dv, dg, K, D):
        n = v.norm(dim=1, keepdim=True)
        vd = (v * dw).sum(1, keepdim=True)
        dv += g / n * (dw - v * vd / n ** 2)
        dg += vd / n

    def colsum_bf16(x, out, rows, N):
        out += x[:rows].float().sum(0)

    def gelu_fwd(x, y, n):
        y.view(-1)[:n] = F.gelu(x.reshape(-1)[:n].float()).to(y.dtype)
        return y

    def gelu_bwd(dy, x, dx, n):
        pre = x.reshape(-1)[:n].float().detach().requires_grad_(True)
        gp, = torch.autograd.grad(F.gelu(pre).sum(), pre)
        dx.view(-1)[:n] = (dy.reshape(-1)[:n].float() * gp).to(dx.dtype)
        return dx

    def im2col_nhwc(x, cols, B, H, W, C, KH, KW, stride, pad):
        u = F.unfold(_rows_to_nchw(x, B, H, W, C), (KH, KW), padding=pad, stride=stride)          # [B, C*KH*KW, L]
        L = u.shape[-1]
        cols[: B * L, : KH * KW * C] = u.view(B, C, KH * KW, L).permute(0, 3, 2, 1).reshape(B * L, KH * KW * C).to(cols.dtype)
        return cols

    def col2im_nhwc(dcols, dx, B, H, W, C, KH, KW, stride, pad, add=None):
        Ho, Wo = ops.conv_out_size(H, KH, stride, pad), ops.conv_out_size(W, KW, stride, pad)
        d = dcols[: B * Ho * Wo, : KH * KW * C].float().view(B, Ho * Wo, KH * KW, C).permute(0, 3, 2, 1).reshape(B, C * KH * KW, Ho * Wo)
        r = _nchw_to_rows(F.fold(d, (H, W), (KH, KW), padding=pad, stride=stride))
        if add is not None:
            r = r + add[: B * H * W].float()
        dx[: B * H * W] = r.to(dx.dtype)
        return dx

    def im2col_nchw_f32(img, cols, KH, KW, stride, pad):
        u = F.unfold(img, (KH, KW), padding=pad, stride=stride).permute(0, 2, 1)
        cols.zero_()
        cols[: u.shape[0] * u.shape[1], : u.shape[2]] = u.reshape(-1, u.shape[2]).to(cols.dtype)
        return cols

    def batchnorm_fwd(x, gamma, beta, y, mean, rstd, rows, C, ws, resid=None, running_mean=None, running_var=None, eps=1e-5, momentum=0.1, relu=False, sync=None):
        assert sync is None, "the emulation covers per-process statistics only"
        xf = x[:rows].float()
        m, v = xf.mean(0), xf.var(0, unbiased=False)
        mean.copy_(m); rstd.copy_((v + eps).rsqrt())
        o = (xf - m) * rstd * gamma + beta
        if resid is not None:
            o = o + resid[:rows].float()
        y[:rows] = (o.relu() if relu else o).to(y.dtype)
        if running_mean is not None:
            running_mean.mul_(1 - momentum).add_(momentum * m)
            running_var.mul_(1 - momentum).add_(momentum * v * rows / max(rows - 1, 1))
        return y

    def batchnorm_apply(x, mean, rstd, gamma, beta, y, rows, C, resid=None, relu=False):
        o = (x[:rows].float() - mean) * rstd * gamma + beta
        if resid is not None:
            o = o + resid[:rows].float()
        y[:rows] = (o.relu() if relu else o).to(y.dtype)
        return y

    def batchnorm_bwd(dy, x, gamma, mean, rstd, dx, rows, C, ws, y=None, dz=None, dgamma=None, dbeta=None, sync=None):
        assert sync is None
        d = dy[:rows].float()
        if y is not None:
            d = d * (y[:rows].float() > 0)
            dz[:rows] = d.to(dz.dtype)
        xh = (x[:rows].float() - mean) * rstd
        if dgamma is not None:
            dgamma += (d * xh).sum(0)
        if dbeta is not None:
            dbeta += d.sum(0)
        dx[:rows] = (gamma * rstd * (d - d.mean(0) - xh * (d * xh).mean(0))).to(dx.dtype)
        return dx

    def maxpool_fwd(x, y, idx, B, H, W, C):
        xin = _rows_to_nchw(x, B, H, W, C).clone().requires_grad_(True)
        o = F.max_pool2d(xin, 3, 2, 1)
        y[: B * o.shape[2] * o.shape[3]] = _nchw_to_rows(o.detach()).to(y.dtype)
        idx._emu = (xin, o)          # test-only side channel: the pooled graph (overlapping windows accumulate in its backward)

    def maxpool_bwd(dy, idx, dx, B, H, W, C):
        xin, o = idx._emu
        d = dy[: o.shape[0] * o.shape[2] * o.shape[3]].float().view(o.shape[0], o.shape[2], o.shape[3], C).permute(0, 3, 1, 2)
        (gx,) = torch.autograd.grad(o, xin, d)
        dx[: B * H * W] = _nchw_to_rows(gx).to(dx.dtype)

    def cast_pad_rows(w, out, D, kreal, kpad):
        out.zero_()
        out[:, :kreal] = w.to(out.dtype)

    def unpad_accumulate(src, dst, D, kreal, kpad):
        dst += src[:, :kreal]

    def add_bf16(a, b, out):
        out.copy_((a.float() + b.float()).to(out.dtype))
        return out

    def token_mean(x, out, B, n, C):
        out.copy_(x[: B * n].float().view(B, n, C).mean(1).to(out.dtype))
        return out

    def pool_bwd_add(d_tok, d_pool, out, B, n, C):
        out[: B * n] = (d_tok[: B * n].view(B, n, C) + d_pool[:, None] / n).reshape(B * n, C).to(out.dtype)
        return out

    for name, fn in (("gemm", gemm), ("im2col_nhwc", im2col_nhwc), ("col2im_nhwc", col2im_nhwc), ("im2col_nchw_f32", im2col_nchw_f32),
                     ("batchnorm_fwd", batchnorm_fwd), ("batchnorm_apply", batchnorm_apply), ("batchnorm_bwd", batchnorm_bwd),
                     ("maxpool3x3s2_fwd", maxpool_fwd), ("maxpool3x3s2_bwd", maxpool_bwd), ("cast_pad_rows", cast_pad_rows),
                     ("unpad_accumulate", unpad_accumulate), ("add_bf16", add_bf16), ("token_mean", token_mean), ("pool_bwd_add", pool_bwd_add),
                     ("batchnorm_ws_floats", lambda C: 8), ("l2norm_fwd", l2norm_fwd), ("l2norm_bwd", l2norm_bwd),
                     ("weightnorm_fwd", weightnorm_fwd), ("weightnorm_bwd", weightnorm_bwd), ("colsum_bf16", colsum_bf16), ("gelu_fwd", gelu_fwd),
                     ("gelu_bwd", gelu_bwd)):
        patch(name, fn)
    try:
        yield
    finally:
        for k, v in saved.items():
            setattr(ops, k, v)
