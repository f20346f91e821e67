"""K-loop rate of the four-phase GEMM on loop-dominated shapes (one round of 256 tiles, K = 8192) and on the step's fc1 shape, for the library in LT_AMD_LIB:
used to compare diagnostic / experimental builds of gemm.hip (lightly_train_amd/build.py::build_variant)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lightly_train_amd  # noqa
from lightly_train_amd import ops
dev = "cuda"
def bench(M, N, K, tb=False, iters=10):
    A = torch.randn(M, K, device=dev).to(torch.bfloat16); B = torch.randn((K, N) if tb else (N, K), device=dev).to(torch.bfloat16)
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    f = lambda: ops.gemm(A, B, C, M=M, N=N, K=K, trans_b=tb, epilogue=ops.EPI_BF16)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for name, M, N, K, tb in (("NN K=8192", 4096, 4096, 8192, False), ("NT(tb) K=8192", 4096, 4096, 8192, True), ("NN K=768 x10 rounds", 50432, 3072, 768, False), ("NT K=3072 N=768", 50432, 768, 3072, True)):
    t = bench(M, N, K, tb)
    print(f"{os.environ.get('LT_AMD_LIB', 'shipped').split('/')[-1]:26s} {name:22s}: {t:8.1f} us  {2 * M * N * K / t / 1e6:7.1f} TF/s")
