"""The row-owning residual GEMM with the fused LayerNorm (lt_gemm_resid_ln768) against the shipped pair (lt_gemm_bf16 with the residual epilogue +
lt_layernorm_fwd): values, then times on the step's two shapes (attention projection K = 768, fc2 K = 3072; M = 50 432 rows)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lightly_train_amd  # noqa
from lightly_train_amd import ops
dev, D = "cuda", 768
g = torch.Generator().manual_seed(0)

def case(M, K, check=True, iters=20):
    A = (torch.randn(M, K, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    W = (torch.randn(D, K, generator=g) * 0.05).to(torch.bfloat16).to(dev)
    bias, gamma = torch.randn(D, generator=g).to(dev) * 0.1, (torch.rand(D, generator=g) + 0.5).to(dev)
    resid = torch.randn(M, D, generator=g).to(dev)
    lw, lb = (torch.rand(D, generator=g) + 0.5).to(dev), (torch.randn(D, generator=g) * 0.1).to(dev)
    out0, out1 = torch.empty(M, D, device=dev), torch.empty(M, D, device=dev)
    y0, y1 = torch.empty(M, D, device=dev, dtype=torch.bfloat16), torch.empty(M, D, device=dev, dtype=torch.bfloat16)
    m0, r0, m1, r1 = (torch.empty(M, device=dev) for _ in range(4))
    def old():
        ops.gemm(A, W, out0, M=M, N=D, K=K, epilogue=ops.EPI_RESID, bias=bias, gamma=gamma, resid=resid)
        ops.layernorm_fwd(out0, lw, lb, M, D, y_bf16=y0, mean=m0, rstd=r0, eps=1e-6)
    def old_gemm():
        ops.gemm(A, W, out0, M=M, N=D, K=K, epilogue=ops.EPI_RESID, bias=bias, gamma=gamma, resid=resid)
    def new():
        ops.gemm_resid_ln768(A, W, out1, M=M, K=K, bias=bias, gamma=gamma, resid=resid, ln_w=lw, ln_b=lb, eps=1e-6, ln_out=y1, mean=m1, rstd=r1)
    def new_gemm():
        ops.gemm_resid_ln768(A, W, out1, M=M, K=K, bias=bias, gamma=gamma, resid=resid)
    old(); new(); torch.cuda.synchronize()
    if check:
        e_out = (out1 - out0).abs().max().item() / out0.abs().max().item()
        e_y = (y1.float() - y0.float()).abs().max().item() / y0.float().abs().max().item()
        e_m, e_r = (m1 - m0).abs().max().item(), ((r1 - r0).abs() / r0).max().item()
        print(f"M={M} K={K}: out rel err {e_out:.2e}  ln_out rel err {e_y:.2e}  mean abs err {e_m:.2e}  rstd rel err {e_r:.2e}")
        assert e_out < 1e-5 and e_y < 1e-2 and e_m < 1e-5 and e_r < 1e-5
    def t(fn):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e3
    print(f"M={M} K={K}: shipped GEMM {t(old_gemm):7.1f} us, + LayerNorm {t(old):7.1f} us | row-owning GEMM alone {t(new_gemm):7.1f} us, with the fused LayerNorm {t(new):7.1f} us")

case(300, 64); case(1000, 768); case(129, 3072)
case(256 * 197, 768, check=True); case(256 * 197, 3072, check=False); case(1024 * 50, 768, check=False)
