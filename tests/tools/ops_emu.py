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
        nb = kw.pop("batch", 1) or 1
        if nb > 1:   # independent problems of one shape, operands stride_* elements apart
            sa, sb, sc = kw.pop("stride_a", 0), kw.pop("stride_b", 0), kw.pop("stride_c", 0)
            la, lb, lc = lda or (M if trans_a else K), ldb or (N if trans_b else K), ldc or N
            ra, rb = (K if trans_a else M), (K if trans_b else N)
            af, bf_, of = a.reshape(-1), b.reshape(-1), out.reshape(-1)
            for i in range(nb):
                gemm(af[i * sa: i * sa + ra * la], bf_[i * sb: i * sb + rb * lb], of[i * sc: i * sc + M * lc], M=M, N=N, K=K, trans_a=trans_a, trans_b=trans_b,
                     epilogue=epilogue, bias=bias, lda=lda, ldb=ldb, ldc=ldc, **kw)
            return out
        for k_ in ("stride_a", "stride_b", "stride_c"):
            kw.pop(k_, None)
        A = (a.float().reshape(-1, lda or (M if trans_a else K))[:K, :M].t() if trans_a else a.float().reshape(-1, lda or K)[:M, :K])
        if kw.get("colsum") is not None:               # bias gradient beside the weight gradient: column sums of dY = row sums of A^T
            assert trans_a and epilogue == ops.EPI_F32_ACCUM
            kw["colsum"].view(-1)[:M] += A.sum(1)
        Bm = (b.float().reshape(-1, ldb or (N if trans_b else K))[:K, :N] if trans_b else b.float().reshape(-1, ldb or K)[:N, :K].t())
        r = A @ Bm
        if kw.get("alpha", 1.0) != 1.0:                # C = alpha * acc + bias (include/lt_amd.h)
            r = r * kw["alpha"]
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
        elif epilogue == ops.EPI_RESID:                # C = resid + branch_scale * rowscale[m] * gamma * y, C2 = y
            if kw.get("out2") is not None:
                kw["out2"].reshape(-1, N)[:M] = r.to(kw["out2"].dtype)
            br = r * (kw["gamma"] if kw.get("gamma") is not None else 1.0)
            if kw.get("rowscale") is not None:
                br = br * kw["rowscale"][:M, None]
            bs = kw.get("branch_scale", 1.0) or 1.0
            base = kw["resid"].reshape(-1, N)[:M].float() if kw.get("resid") is not None else 0.0
            o2[:M, :N] = (base + bs * br).to(out.dtype)
            if kw.get("ln") is not None:               # the next LayerNorm behind the residual GEMM (lt_gemm_desc.ln_*): GEMM + lt_layernorm_fwd
                ln = kw["ln"]
                layernorm_fwd(o2[:M, :N], ln["weight"], ln["bias"], M, N, y_bf16=ln["out"], mean=ln.get("mean"), rstd=ln.get("rstd"), eps=ln["eps"])
        else:
            assert epilogue in (ops.EPI_BF16, ops.EPI_F32), epilogue
            o2[:M, :N] = r.to(out.dtype)
        return out

    # ---- token path / norms / attention of the ViT engine (semantics: include/lt_amd.h) ----
    def matmul_f32(a, b, out, M, N, K, trans_a=False, accumulate=False):
        A = a.reshape(K, M).t() if trans_a else a.reshape(M, K)
        r = A @ b.reshape(K, N)
        if accumulate:
            out.view(M, N).add_(r)
        else:
            out.view(M, N).copy_(r)
        return out

    def resize_4tap(img, iy, wy, ix, wx, Ho, Wo):
        t = (img[:, :, iy.long(), :] * wy[None, None, :, :, None]).sum(3)                  # [B, C, Ho, W]
        return (t[:, :, :, ix.long()] * wx[None, None, None]).sum(4).contiguous()          # [B, C, Ho, Wo]

    def im2col(img, p_, kpad):
        B, Cc, H, W = img.shape
        u = F.unfold(img, (p_, p_), stride=p_).transpose(1, 2).reshape(-1, Cc * p_ * p_)   # k = (c*p + py)*p + px
        cols = torch.zeros(u.shape[0], kpad, dtype=torch.float32)
        cols[:, : u.shape[1]] = u
        return cols

    def assemble_tokens(patch, cls, pos, mask_token, masks, B, n_p, D, out=None, reg=None, n_reg=0):
        x = out if out is not None else torch.empty(B, n_p + 1 + n_reg, D)
        xv = x.view(B, n_p + 1 + n_reg, D)
        pv = patch.float().reshape(B, n_p, D)
        if masks is not None:
            pv = torch.where(masks.view(B, n_p, 1).bool(), mask_token.view(1, 1, D), pv)
        posv = pos.view(-1, D)
        xv[:, 0] = cls.view(1, D) + posv[0]
        if n_reg:
            xv[:, 1:1 + n_reg] = reg.view(1, n_reg, D)
        xv[:, 1 + n_reg:] = pv + posv[1:]
        return x

    def assemble_tokens_bwd(dx, masks, dpatch, dcls, dpos, dmask, B, n_p, D, dreg=None, n_reg=0):
        d = dx.view(B, n_p + 1 + n_reg, D)
        dp = d[:, 1 + n_reg:]
        dcls.view(-1).add_(d[:, 0].sum(0))
        dpos.view(-1, D)[0].add_(d[:, 0].sum(0))
        dpos.view(-1, D)[1:].add_(dp.sum(0))
        if n_reg:
            dreg.view(n_reg, D).add_(d[:, 1:1 + n_reg].sum(0))
        if masks is not None:
            mk = masks.view(B, n_p, 1).bool()
            dmask.view(-1).add_((dp * mk).sum((0, 1)))
            dp = dp * (~mk)
        dpatch.view(B * n_p, D).copy_(dp.reshape(B * n_p, D).to(dpatch.dtype))

    def layernorm_fwd(x, w, b, rows, D, y_bf16=None, y_f32=None, mean=None, rstd=None, eps=1e-6):
        xr = x.reshape(-1, D)[:rows]
        m = xr.mean(1)
        r = (xr.var(1, unbiased=False) + eps).rsqrt()
        y = (xr - m[:, None]) * r[:, None] * w + b
        if y_bf16 is not None:
            y_bf16.reshape(-1, D)[:rows] = y.to(y_bf16.dtype)
        if y_f32 is not None:
            y_f32.reshape(-1, D)[:rows] = y
        if mean is not None:
            mean.view(-1)[:rows] = m
        if rstd is not None:
            rstd.view(-1)[:rows] = r

    def layernorm_bwd(x, w, mean, rstd, dy, dres, dx, dw, db, rows, D, ws=None, dnext=None, gamma_next=None, rowscale_next=None,
                      scale_next=1.0, dbias_next=None, ridx=None):
        if ridx is not None:      # dx[ridx[r]] = dres[ridx[r]] + LN'(dy[r])
            ii = ridx[:rows].long()
            tmp = torch.zeros(rows, D)
            layernorm_bwd(x, w, mean, rstd, dy, None, tmp, dw, db, rows, D)
            base = dres.reshape(-1, D)[ii] if dres is not None else 0.0
            dx.reshape(-1, D)[ii] = base + tmp
            return
        xr = x.reshape(-1, D)[:rows]
        d = dy.reshape(-1, D)[:rows].float()
        xh = (xr - mean.view(-1)[:rows, None]) * rstd.view(-1)[:rows, None]
        dw += (d * xh).sum(0)
        db += d.sum(0)
        g = d * w
        dxx = (g - g.mean(1, keepdim=True) - xh * (g * xh).mean(1, keepdim=True)) * rstd.view(-1)[:rows, None]
        if dres is not None:
            dxx = dxx + dres.reshape(-1, D)[:rows]
        dx.reshape(-1, D)[:rows] = dxx
        if dnext is not None:
            dn = dxx * (gamma_next if gamma_next is not None else 1.0) * scale_next
            if rowscale_next is not None:
                dn = dn * rowscale_next[:rows, None]
            dnext.reshape(-1, D)[:rows] = dn.to(dnext.dtype)
            if dbias_next is not None:
                dbias_next += dn.sum(0)

    def layerscale_bwd(dout, y, gamma, dy, dgamma, rows, D, dbias=None, rowscale=None, scale=1.0, ridx=None):
        d = dout.reshape(-1, D)[:rows] if ridx is None else dout.reshape(-1, D)[ridx[:rows].long()]
        m = scale * (rowscale[:rows, None] if rowscale is not None else 1.0)
        if gamma is not None and y is not None:
            dgamma += (d * y.reshape(-1, D)[:rows].float() * m).sum(0)
        o = d * (gamma if gamma is not None else 1.0) * m
        dy.reshape(-1, D)[:rows] = o.to(dy.dtype)
        if dbias is not None:
            dbias += o.sum(0)

    def layerscale_dgamma(w, dw, bias, dbias, gamma, dgamma, N, K):
        acc = (w.float().view(N, K) * dw.view(N, K)).sum(1)
        if bias is not None and dbias is not None:
            acc = acc + bias * dbias
        dgamma += torch.where(gamma.abs() > 1e-30, acc / gamma, torch.zeros_like(acc))

    def layerscale_dgamma_batched(w, dw, bias, dbias, gamma, dgamma, N, K, batch, stride):
        for i in range(batch):
            at = lambda t, n: None if t is None else torch.as_strided(t, (n,), (1,), t.storage_offset() + i * stride)   # noqa: E731
            layerscale_dgamma(at(w, N * K), at(dw, N * K), at(bias, N), at(dbias, N), at(gamma, N), at(dgamma, N), N, K)

    def gather_rows(src, ld, idx, M, D, out_bf16=None, out_f32=None):
        r = src.reshape(-1, ld)[idx[:M], :D]
        if out_bf16 is not None:
            out_bf16.reshape(-1, D)[:M] = r.to(out_bf16.dtype)
        if out_f32 is not None:
            out_f32.reshape(-1, D)[:M] = r.float()

    def scatter_add_rows(src, idx, dst, ld, M, D):
        dst.reshape(-1, ld)[:, :D].index_add_(0, idx[:M], src.reshape(-1, D)[:M].to(dst.dtype))

    def _attn(qkv, B, N, Hh, dh, scale):
        q, k, v = qkv.reshape(B, N, 3, Hh, dh).float().permute(2, 0, 3, 1, 4)
        return ((q * scale) @ k.transpose(-2, -1)).softmax(-1) @ v                          # [B, H, N, dh]

    def attention_fwd(qkv, out, lse, B, N, Hh, dh, scale):
        o = _attn(qkv.reshape(-1)[: B * N * 3 * Hh * dh], B, N, Hh, dh, scale)
        out.reshape(-1)[: B * N * Hh * dh] = o.transpose(1, 2).reshape(-1).to(out.dtype)

    def attention_bwd(qkv, out, dout, lse, ws, dqkv, B, N, Hh, dh, scale):
        n = B * N * 3 * Hh * dh
        leaf = qkv.reshape(-1)[:n].float().detach().clone().requires_grad_(True)
        o = _attn(leaf, B, N, Hh, dh, scale).transpose(1, 2).reshape(B, N, Hh * dh)
        g, = torch.autograd.grad(o, leaf, dout.reshape(-1)[: B * N * Hh * dh].float().view(B, N, Hh * dh))
        dqkv.reshape(-1)[:n] = g.to(dqkv.dtype)

    # ---- losses / optimizer of the DINOv2 method (semantics: include/lt_amd.h) ----
    def softmax_center(logits, center, probs, rows, K, inv_temp):
        z = logits.reshape(-1, K)[:rows]
        if center is not None:
            z = z - center.view(1, K)
        probs.reshape(-1, K)[:rows] = torch.softmax(z * inv_temp, dim=-1)

    def center_ema(center, colsum, scale, momentum, K):
        center.view(-1).mul_(momentum).add_(colsum.view(-1) * (scale * (1.0 - momentum)))

    def colsum_f32(x, out, rows, N, accumulate=False):
        sres = x.reshape(-1, N)[:rows].sum(0)
        if accumulate:
            out.view(-1).add_(sres)
        else:
            out.view(-1).copy_(sres)

    def scale_f32(dst, alpha):
        dst.mul_(alpha)

    def fill_f32(dst, value):
        dst.fill_(value)

    def ce_fwd_bwd(s_, teacher, ta, tb, row_weight, scale, inv_temp, loss, dlogits, rows, K, slot=None):
        sl = s_.reshape(-1, K)[:rows]
        t = teacher.reshape(-1, K)[ta[:rows].long()]
        if tb is not None:
            has = (tb[:rows] >= 0)
            t = t + teacher.reshape(-1, K)[tb[:rows].clamp_min(0).long()] * has[:, None]
        coef = scale * (row_weight[:rows] if row_weight is not None else torch.ones(rows))
        lsm = torch.log_softmax(sl * inv_temp, dim=-1)
        l = -(t * lsm).sum(-1) * coef
        if slot is None:
            loss[0] += l.sum()
        else:
            loss.index_add_(0, slot[:rows].long(), l)
        if dlogits is not None:
            dlogits.reshape(-1, K)[:rows] = ((coef * inv_temp)[:, None] * (lsm.exp() * t.sum(-1, keepdim=True) - t)).to(dlogits.dtype)

    def softmax_stats_colsum(logits, center, stats, colsum, rows, K, inv_temp, scratch=None):
        x = logits.reshape(-1, K)[:rows]
        z = (x - center.view(1, K) if center is not None else x) * inv_temp
        m = z.max(-1).values if rows else z.new_zeros(0)
        st = stats.reshape(-1, 2)
        st[:rows, 0] = m
        st[:rows, 1] = 1.0 / torch.exp(z - m[:, None]).sum(-1)
        colsum.view(-1).copy_(x.sum(0))

    def ce_fwd_bwd_logits(s_, t_logits, t_stats, center_a, center_b, split_row, ta, tb, row_weight, scale, inv_temp, inv_temp_t, loss, dlogits,
                          rows, K, slot=None):
        tl = t_logits.reshape(-1, K)
        nt = tl.shape[0]
        cen = torch.zeros(nt, K)
        if center_a is not None:
            cen[:split_row] = center_a.view(1, K)
        if center_b is not None:
            cen[split_row:] = center_b.view(1, K)
        st = t_stats.reshape(-1, 2)
        used = torch.unique(torch.cat([ta[:rows].long(), tb[:rows].clamp_min(0).long()] if tb is not None else [ta[:rows].long()]))
        probs = torch.zeros(nt, K)       # only the rows that are referenced have statistics
        probs[used] = torch.exp((tl[used] - cen[used]) * inv_temp_t - st[used, 0:1]) * st[used, 1:2]
        ce_fwd_bwd(s_, probs, ta, tb, row_weight, scale, inv_temp, loss, dlogits, rows, K, slot=slot)

    def roi_resample_tokens(x, src_image, idx, w, B, img_stride, n_out, D, out_bf16=None, out_f32=None):
        flat = x.reshape(-1)
        for b in range(B):
            sb = int(src_image[b]) if src_image is not None else b
            img = flat[sb * img_stride: sb * img_stride + (int(idx[b].max()) + 1) * D].view(-1, D)
            r = (img[idx[b].long()] * w[b][:, :, None]).sum(1)                      # [n_out, D]
            for o_ in (out_bf16, out_f32):
                if o_ is not None:
                    o_.reshape(-1, D)[b * n_out:(b + 1) * n_out] = r.to(o_.dtype)

    def roi_resample_tokens_bwd(dout, idx, w, din, B, img_stride, n_in, n_out, D):
        flat = din.reshape(-1)
        for b in range(B):
            acc = torch.zeros(n_in, D)
            g_ = dout.reshape(-1, D)[b * n_out:(b + 1) * n_out].float()
            for a_ in range(4):
                acc.index_add_(0, idx[b][:, a_].long(), g_ * w[b][:, a_:a_ + 1])
            flat[b * img_stride: b * img_stride + n_in * D] = acc.reshape(-1)

    def center_tokens(z, B, n, C, out_bf16=None, out_f32=None):
        zz = z.reshape(-1, C)[: B * n].view(B, n, C)
        r = (zz - zz.mean(1, keepdim=True)).reshape(B * n, C)
        for o_ in (out_bf16, out_f32):
            if o_ is not None:
                o_.reshape(-1, C)[: B * n] = r.to(o_.dtype)

    def cka_fwd_bwd(Ks, Kt, coef, loss, G, B, n, ld, eps=1e-8):
        ks = Ks.reshape(-1, ld)[: B * n].view(B, n, ld)[:, :, :n].float()
        kt = Kt.reshape(-1, ld)[: B * n].view(B, n, ld)[:, :, :n].float()
        hst, ns, nt = (ks * kt).sum((1, 2)), ks.pow(2).sum((1, 2)).sqrt(), kt.pow(2).sum((1, 2)).sqrt()
        den = ns * nt + eps
        loss[0] += (coef[:B] * (1.0 - hst / den)).sum()
        if G is not None:
            gs = torch.where(ns > 0, coef[:B] * hst * nt / (ns * den * den), torch.zeros_like(ns))
            g = (-coef[:B] / den)[:, None, None] * kt + gs[:, None, None] * ks
            Gv = G.reshape(-1, ld)[: B * n].view(B, n, ld)
            Gv.zero_()
            Gv[:, :, :n] = g.to(G.dtype)

    def sk_exp(logits, Q, inv_temp):
        Q.reshape(-1)[: logits.numel()] = torch.exp(logits.reshape(-1) * inv_temp)

    def sk_iter(Q, colsum, rows, K, n_total, final_mul):
        q = Q.reshape(-1, K)[:rows]
        q = q / (colsum.view(1, K) * K)
        q = q / (q.sum(-1, keepdim=True) * n_total)
        Q.reshape(-1, K)[:rows] = q * final_mul

    def koleo_fwd_bwd(x, ld, loss, dx, ld_dx, n, D, weight, ws, nn, eps=1e-8):
        from oracle import dinov2_oracle as O_
        rows = torch.as_strided(x, (n, D), (ld, 1)).detach().clone().requires_grad_(True)
        L = O_.koleo_loss(rows, eps=eps)
        if weight == 0.0:
            loss[0] += L.detach()
            return
        g, = torch.autograd.grad(L, rows)
        loss[0] += weight * L.detach()
        torch.as_strided(dx, (n, D), (ld_dx, 1)).add_(weight * g)

    def sumsq(g, out):
        out[0] += (g.double() ** 2).sum().float()

    def adamw_flat(p_, g, m, v, p_bf16, seg_of_chunk, seg_lr, seg_wd_on, seg_frozen, freeze, lr_factor, wd, beta1, beta2, eps, step, sumsq_t, max_norm):
        seg = seg_of_chunk.long().repeat_interleave(1024)
        lr = seg_lr[seg] * lr_factor
        lr = torch.where((seg_frozen[seg].int() & int(freeze)) != 0, torch.zeros_like(lr), lr)
        wdv = torch.where(seg_wd_on[seg] != 0, torch.full_like(lr, wd), torch.zeros_like(lr))
        clip = 1.0
        if max_norm > 0:
            clip = min(1.0, max_norm / (float(sumsq_t[0]) ** 0.5 + 1e-6))
        gr = g * clip
        p_.mul_(1 - lr * wdv)
        m.add_((gr - m) * (1 - beta1))
        v.mul_(beta2).add_(gr * gr * (1 - beta2))
        bc1, bc2s = 1 - beta1 ** step, (1 - beta2 ** step) ** 0.5
        p_.sub_(lr / bc1 * (m / (v.sqrt() / bc2s + eps)))
        if p_bf16 is not None:
            p_bf16.copy_(p_.to(p_bf16.dtype))

    def ema_flat(teacher, student, teacher_bf16, m_):
        teacher.mul_(m_).add_(student * (1.0 - m_))
        if teacher_bf16 is not None:
            teacher_bf16.copy_(teacher.to(teacher_bf16.dtype))

    # ---- distillation methods ----
    def resample_tokens(x, idx, w, out, B, n_in, n_out, D, taps):
        xv = x.reshape(-1)[: B * n_in * D].view(B, n_in, D)
        o = (xv[:, idx.long().view(-1)].view(B, n_out, taps, D) * w.view(1, n_out, taps, 1)).sum(2)
        out.reshape(-1)[: B * n_out * D] = o.reshape(-1)

    def kl_fwd_bwd(s_logits, t_logits, ld, inv_temp, coef, loss, dlogits, ldd, rows, K):
        sv = torch.as_strided(s_logits, (rows, K), (ld, 1)); tv = torch.as_strided(t_logits, (rows, K), (ld, 1))
        ls, lt = torch.log_softmax(sv * inv_temp, -1), torch.log_softmax(tv * inv_temp, -1)
        loss[0] += coef * (lt.exp() * (lt - ls)).sum()
        if dlogits is not None:
            torch.as_strided(dlogits, (rows, K), (ldd, 1)).copy_((coef * inv_temp * (ls.exp() - lt.exp())).to(dlogits.dtype))

    def cast_bf16(src, dst):
        dst.reshape(-1)[: src.numel()] = src.reshape(-1).to(dst.dtype)

    def symmetrize_bf16(d, g, batch, n, ld):
        dv = torch.as_strided(d, (batch, n, n), (n * ld, ld, 1)).float()
        torch.as_strided(g, (batch, n, n), (n * ld, ld, 1)).copy_((dv + dv.transpose(1, 2)).to(g.dtype))

    def mixup(x, index, lam, out):
        out.copy_(lam * x + (1.0 - lam) * x[index])

    def mse_fwd_bwd(s_, t, ds, n, scale, loss):
        d = s_.reshape(-1)[:n] - t.reshape(-1)[:n]
        loss[0] += scale * (d * d).sum()
        if ds is not None:
            ds.reshape(-1)[:n] = 2.0 * scale * d

    def lars_flat(p_, g, buf, p_bf16, seg_of_chunk, seg_chunk_begin, seg_lr, seg_wd_on, ws, seg_norms, lr_factor, wd, momentum, dampening, nesterov,
                  trust, eps, first_step, sumsq_t, max_norm):
        clip = 1.0
        if max_norm > 0:
            clip = min(1.0, max_norm / (float(sumsq_t[0]) ** 0.5 + 1e-6))
        cb = seg_chunk_begin.tolist()
        for si in range(len(cb) - 1):
            sl = slice(cb[si] * 1024, cb[si + 1] * 1024)
            pp, gg = p_[sl], g[sl] * clip
            wdv = wd if int(seg_wd_on[si]) else 0.0
            pn, gn = float(pp.norm()), float(gg.norm())
            d = gg
            if wdv != 0 and pn != 0 and gn != 0:
                d = (gg + wdv * pp) * (pn / (gn + pn * wdv + eps) * trust)
            if momentum != 0:
                if first_step:
                    buf[sl] = d
                else:
                    buf[sl] = buf[sl] * momentum + (1 - dampening) * d
                d = d + momentum * buf[sl] if nesterov else buf[sl]
            pp.sub_(float(seg_lr[si]) * lr_factor * d)
        if p_bf16 is not None:
            p_bf16.copy_(p_.to(p_bf16.dtype))

    def sgd_flat(p_, g, buf, p_bf16, seg_of_chunk, seg_lr, seg_wd_on, lr_factor, wd, momentum, dampening, nesterov, first_step, sumsq_t, max_norm):
        clip = 1.0
        if max_norm > 0:
            clip = min(1.0, max_norm / (float(sumsq_t[0]) ** 0.5 + 1e-6))
        seg = seg_of_chunk.long().repeat_interleave(1024)
        lr = seg_lr[seg] * lr_factor
        d = g * clip + (seg_wd_on[seg].float() * wd) * p_
        if momentum != 0:
            if first_step:
                buf.copy_(d)
            else:
                buf.mul_(momentum).add_(d, alpha=1 - dampening)
            d = d + momentum * buf if nesterov else buf
        p_.sub_(lr * d)
        if p_bf16 is not None:
            p_bf16.copy_(p_.to(p_bf16.dtype))

    def rope_apply(qkv, sin_t, cos_t, B, N, Hh, dh, prefix, inverse=False):
        v = qkv.reshape(-1)[: B * N * 3 * Hh * dh].view(B, N, 3, Hh, dh)
        x = v[:, prefix:, :2].float()                                    # q and k of the patch tokens
        s_, c_ = sin_t.view(1, -1, 1, 1, dh), cos_t.view(1, -1, 1, 1, dh)
        x1, x2 = x[..., : dh // 2], x[..., dh // 2:]
        if not inverse:
            r = x * c_ + torch.cat([-x2, x1], -1) * s_                   # x cos + rotate_half(x) sin
        else:                                                            # transposed rotation
            y = x * s_
            r = x * c_ + torch.cat([y[..., dh // 2:], -y[..., : dh // 2]], -1)
        v[:, prefix:, :2] = r.to(v.dtype)

    def swiglu_fwd(x12, out, rows, Hd):
        a = x12.reshape(-1, 2 * Hd)[:rows].float()
        out.reshape(-1, Hd)[:rows] = (F.silu(a[:, :Hd]) * a[:, Hd:]).to(out.dtype)

    def swiglu_bwd(x12, dh_, d12, rows, Hd):
        a = x12.reshape(-1, 2 * Hd)[:rows].float().detach().clone().requires_grad_(True)
        o = F.silu(a[:, :Hd]) * a[:, Hd:]
        g, = torch.autograd.grad(o, a, dh_.reshape(-1, Hd)[:rows].float())
        d12.reshape(-1, 2 * Hd)[:rows] = g.to(d12.dtype)

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
        xf = x[:rows].float()
        if sync is None:
            m, v, n = xf.mean(0), xf.var(0, unbiased=False), rows
        else:   # SyncBatchNorm: (sum x, sum x^2, rows) summed over the ranks by the caller's all-reduce
            sums = torch.cat([xf.double().sum(0), (xf.double() ** 2).sum(0), torch.tensor([float(rows)], dtype=torch.float64)])
            sync(sums)
            n = float(sums[-1])
            m = (sums[:C] / n).float()
            v = (sums[C:2 * C] / n - (sums[:C] / n) ** 2).clamp_min(0).float()
            if rows == 0:
                return y
        mean.copy_(m); rstd.copy_((v + eps).rsqrt())
        o = (xf - m) * rstd * gamma + beta
        if resid is not None:
            o = o + resid[:rows].float()
        y[:rows] = (o.relu() if relu else o).to(y.dtype)
        if running_mean is not None:
            running_mean.mul_(1 - momentum).add_(momentum * m)
            running_var.mul_(1 - momentum).add_(momentum * v * n / max(n - 1, 1))
        return y

    def batchnorm_apply(x, mean, rstd, gamma, beta, y, rows, C, resid=None, relu=False):
        o = (x[:rows].float() - mean) * rstd * gamma + beta
        if resid is not None:
            o = o + resid[:rows].float()
        y[:rows] = (o.relu() if relu else o).to(y.dtype)
        return y

    def batchnorm_bwd(dy, x, gamma, mean, rstd, dx, rows, C, ws, y=None, dz=None, dgamma=None, dbeta=None, sync=None):
        d = dy[:rows].float()
        if y is not None:
            d = d * (y[:rows].float() > 0)
            dz[:rows] = d.to(dz.dtype)
        xh = (x[:rows].float() - mean) * rstd
        if dgamma is not None:
            dgamma += (d * xh).sum(0)          # parameter gradients: local sums (the data-parallel mean handles them)
        if dbeta is not None:
            dbeta += d.sum(0)
        if sync is None:
            c1, c2 = d.mean(0), (d * xh).mean(0)
        else:
            sums = torch.cat([d.double().sum(0), (d * xh).double().sum(0), torch.tensor([float(rows)], dtype=torch.float64)])
            sync(sums)
            c1, c2 = (sums[:C] / sums[-1]).float(), (sums[C:2 * C] / sums[-1]).float()
            if rows == 0:
                return dx
        dx[:rows] = (gamma * rstd * (d - c1 - xh * c2)).to(dx.dtype)
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
        t = d_tok[: B * n].view(B, n, C) if d_tok is not None else torch.zeros(B, n, C)     # either operand may be absent (lt_pool_bwd_add)
        if d_pool is not None:
            t = t + d_pool[:B, None] / n
        out[: B * n] = t.reshape(B * n, C).to(out.dtype)
        return out

    for name, fn in (("gemm", gemm), ("im2col_nhwc", im2col_nhwc), ("col2im_nhwc", col2im_nhwc), ("im2col_nchw_f32", im2col_nchw_f32),
                     ("batchnorm_fwd", batchnorm_fwd), ("batchnorm_apply", batchnorm_apply), ("batchnorm_bwd", batchnorm_bwd),
                     ("maxpool3x3s2_fwd", maxpool_fwd), ("maxpool3x3s2_bwd", maxpool_bwd), ("cast_pad_rows", cast_pad_rows),
                     ("unpad_accumulate", unpad_accumulate), ("add_bf16", add_bf16), ("token_mean", token_mean), ("pool_bwd_add", pool_bwd_add),
                     ("batchnorm_ws_floats", lambda C: 8), ("l2norm_fwd", l2norm_fwd), ("l2norm_bwd", l2norm_bwd),
                     ("weightnorm_fwd", weightnorm_fwd), ("weightnorm_bwd", weightnorm_bwd), ("colsum_bf16", colsum_bf16), ("gelu_fwd", gelu_fwd),
                     ("gelu_bwd", gelu_bwd), ("matmul_f32", matmul_f32), ("resize_4tap", resize_4tap), ("im2col", im2col),
                     ("assemble_tokens", assemble_tokens), ("assemble_tokens_bwd", assemble_tokens_bwd), ("layernorm_fwd", layernorm_fwd),
                     ("layernorm_bwd", layernorm_bwd), ("layerscale_bwd", layerscale_bwd), ("layerscale_dgamma", layerscale_dgamma), ("layerscale_dgamma_batched", layerscale_dgamma_batched),
                     ("gather_rows", gather_rows), ("scatter_add_rows", scatter_add_rows), ("attention_fwd", attention_fwd),
                     ("attention_bwd", attention_bwd), ("attention_bwd_ws_floats", lambda B, N, H, dh: 8), ("swiglu_fwd", swiglu_fwd),
                     ("swiglu_bwd", swiglu_bwd), ("softmax_center", softmax_center), ("roi_resample_tokens", roi_resample_tokens), ("roi_resample_tokens_bwd", roi_resample_tokens_bwd), ("center_tokens", center_tokens), ("cka_fwd_bwd", cka_fwd_bwd), ("softmax_stats_colsum", softmax_stats_colsum), ("ce_fwd_bwd_logits", ce_fwd_bwd_logits), ("center_ema", center_ema), ("colsum_f32", colsum_f32),
                     ("scale_f32", scale_f32), ("fill_f32", fill_f32), ("ce_fwd_bwd", ce_fwd_bwd), ("sk_exp", sk_exp), ("sk_iter", sk_iter),
                     ("koleo_fwd_bwd", koleo_fwd_bwd), ("sumsq", sumsq), ("adamw_flat", adamw_flat), ("ema_flat", ema_flat),
                     ("resample_tokens", resample_tokens), ("kl_fwd_bwd", kl_fwd_bwd), ("cast_bf16", cast_bf16), ("symmetrize_bf16", symmetrize_bf16),
                     ("mixup", mixup), ("mse_fwd_bwd", mse_fwd_bwd), ("lars_flat", lars_flat), ("sgd_flat", sgd_flat), ("rope_apply", rope_apply),
                     ("require_device", lambda dev, who: None)):
        patch(name, fn)
    try:
        yield
    finally:
        for k, v in saved.items():
            setattr(ops, k, v)
