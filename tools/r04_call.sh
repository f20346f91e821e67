#!/bin/bash
# round-4 GPU call: tests of the new kernels / paths, in-step A/B of their switches, default bench line
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r04a}
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_ops_contract.py -x -q -m gpu > $O/tests_ops.log 2>&1
tail -3 $O/tests_ops.log
timeout 1200 python -m pytest tests/test_gpu_step.py -x -q -m gpu > $O/tests_step.log 2>&1
tail -3 $O/tests_step.log
for b2 in 0 1; do echo "LT_ATTN_FWD_B2=$b2"; LT_ATTN_FWD_B2=$b2 timeout 120 python tools/attn_bench.py 2>&1 | head -3; done > $O/attn_fwd_b2.log 2>&1; cat $O/attn_fwd_b2.log
timeout 300 python tools/ab_step.py LT_ATTN_FWD_B2 0 1 --steps 25 > $O/ab_attn_fwd_b2.log 2>&1; tail -2 $O/ab_attn_fwd_b2.log
timeout 300 python tools/ab_step.py LT_GEMM_WGRAD_SLICES 3 4 64 --steps 20 > $O/ab_wgrad_slices.log 2>&1; tail -3 $O/ab_wgrad_slices.log
timeout 300 python tools/ab_step.py fused_centering 0 1 --attr --steps 25 > $O/ab_fused_centering.log 2>&1; tail -2 $O/ab_fused_centering.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log
