"""Run one GEMM shape a few times (for rocprofv3 --pmc passes)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lightly_train_amd
from lightly_train_amd import ops
M, N, K, tb, fk = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
B = (torch.randn(K, N, device="cuda") if tb else torch.randn(N, K, device="cuda")).to(torch.bfloat16)
C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for _ in range(5):
    ops.gemm(A, B, C, M=M, N=N, K=K, trans_b=bool(tb), epilogue=ops.EPI_BF16, force_kernel=fk)
torch.cuda.synchronize()
