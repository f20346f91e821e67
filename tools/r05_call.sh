#!/bin/bash
R=$GRAFT_REPO_ROOT
T=${1:-r05m}
O=$R/gpurun_out/$T
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py -q -k "attention" > $O/t_attn.log 2>&1; tail -4 $O/t_attn.log | cut -c1-200
python - <<'PY' 2>&1 | tee $O/attn_bwd_257.log
import os, sys, torch
sys.path.insert(0, os.getcwd())
import lightly_train_amd
from lightly_train_amd import ops
def bench(B, N, H=12, dh=64, iters=10):
    qkv = torch.randn(B, N, 3 * H * dh, device="cuda").to(torch.bfloat16)
    out = torch.empty(B, N, H * dh, device="cuda", dtype=torch.bfloat16); lse = torch.empty(B, H, N, device="cuda")
    dout = torch.randn(B, N, H * dh, device="cuda").to(torch.bfloat16)
    ws = torch.empty(ops.attention_bwd_ws_floats(B, N, H, dh), device="cuda"); dqkv = torch.empty_like(qkv)
    ops.attention_fwd(qkv, out, lse, B, N, H, dh, dh ** -0.5)
    fn = lambda: ops.attention_bwd(qkv, out, dout, lse, ws, dqkv, B, N, H, dh, dh ** -0.5)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
print("attention backward, B = 256, H = 12, head_dim 64 (us): fused two-pass | 8-wave dQ + dK/dV pair | 4-wave pair")
for N in (225, 257, 261, 288):
    os.environ["LT_ATTN_BWD_2P"] = "1"; a = bench(256, N)
    os.environ["LT_ATTN_BWD_2P"] = "0"; os.environ["LT_ATTN_BWD_V2_ALL"] = "1"; b = bench(256, N)
    os.environ["LT_ATTN_BWD_V2_ALL"] = "0"; c = bench(256, N)
    del os.environ["LT_ATTN_BWD_2P"], os.environ["LT_ATTN_BWD_V2_ALL"]
    print(f"N={N}: {a:7.1f} | {b:7.1f} | {c:7.1f}")
PY
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --patch-size 14 > $O/bench_p14.log 2>&1; tail -1 $O/bench_p14.log | cut -c1-200
