"""The LayerScale + fp32-residual GEMMs of the step (attention projection K = 768, fc2 K = 3072; global / local token counts) for the library in LT_AMD_LIB:
compares epilogue variants of gemm.hip (lightly_train_amd/build.py::build_variant)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lightly_train_amd  # noqa
from lightly_train_amd import ops
dev = "cuda"
def t_of(f, iters=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
g = torch.Generator().manual_seed(0)
name = os.environ.get("LT_AMD_LIB", "shipped").split("/")[-1]
sums = []
for M in (50432, 37888):
    for K in (768, 3072):
        N = 768
        A = torch.randn(M, K, generator=g).to(dev).to(torch.bfloat16); W = (torch.randn(N, K, generator=g) * 0.02).to(dev).to(torch.bfloat16)
        bias, gamma = torch.randn(N, generator=g).to(dev), torch.rand(N, generator=g).to(dev)
        # round-robin over 4 residual / output pairs: a 155 MB stream that was just written is otherwise served from the Infinity Cache
        rs = [torch.randn(M, N, generator=g).to(dev) for _ in range(4)]; outs = [torch.empty(M, N, device=dev) for _ in range(4)]
        i = [0]
        def f():
            j = i[0] % 4; i[0] += 1
            ops.gemm(A, W, outs[j], M=M, N=N, K=K, epilogue=ops.EPI_RESID, bias=bias, gamma=gamma, resid=rs[j])
        t = t_of(f)
        sums.append(float(outs[0].double().sum()))
        print(f"{name:24s} M {M} K {K:5d}: {t:7.1f} us   checksum {sums[-1]:.6f}")
