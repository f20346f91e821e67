import os, sys, torch
sys.path.insert(0, "/root/repo")
import lightly_train_amd
from lightly_train_amd import ops
T, D = 50432, 768
dev = "cuda"
A = torch.randn(T, D, device=dev).to(torch.bfloat16)
B = torch.randn(4 * D, D, device=dev).to(torch.bfloat16)
C = torch.empty(T, 4 * D, device=dev, dtype=torch.bfloat16)
C2 = torch.empty_like(C)
bias = torch.zeros(4 * D, device=dev)
def run(name, **kw):
    for _ in range(2): ops.gemm(A, B, C, M=T, N=4 * D, K=D, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.gemm(A, B, C, M=T, N=4 * D, K=D, **kw)
    e1.record(); torch.cuda.synchronize()
    print(f"{name:40s} {e0.elapsed_time(e1)/10*1e3:8.1f} us")
run("plain bf16", epilogue=ops.EPI_BF16, bias=bias)
run("gelu, no pre store", epilogue=ops.EPI_BF16_GELU, bias=bias)
run("gelu + pre store", epilogue=ops.EPI_BF16_GELU, bias=bias, out2=C2)
