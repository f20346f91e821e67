"""Keep the fc1 forward GEMM (+GELU) running for ~15 s (workload for tools/clock_probe.sh)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lightly_train_amd  # noqa: F401
from lightly_train_amd import ops
T, D = 256 * 197, 768
A = torch.randn(T, D, device="cuda").to(torch.bfloat16); B = torch.randn(4 * D, D, device="cuda").to(torch.bfloat16)
C = torch.empty(T, 4 * D, device="cuda", dtype=torch.bfloat16); C2 = torch.empty_like(C); bias = torch.zeros(4 * D, device="cuda")
t0 = time.time()
while time.time() - t0 < 14:
    for _ in range(200):
        ops.gemm(A, B, C, M=T, N=4 * D, K=D, epilogue=ops.EPI_BF16_GELU, bias=bias, out2=C2)
    torch.cuda.synchronize()
