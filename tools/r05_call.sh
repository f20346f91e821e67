#!/bin/bash
# Scratch call script of round 5 (rewritten per gpurun call).
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/r05_call.sh <tag>'
R=$GRAFT_REPO_ROOT
T=${1:-r05a}
O=$R/gpurun_out/$T
mkdir -p $O
cd $R
export TMPDIR=/tmp
# 1. correctness of what changed: attention (new fused local-crop backward), the two bench-configuration fixtures, the ViT-S-width trajectory
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "attention" > $O/t_attn.log 2>&1; tail -2 $O/t_attn.log
timeout 900 python -m pytest tests/test_gpu_step.py -q -x -k "bench_configuration or vits_width or matches_reference_fixture" > $O/t_step.log 2>&1; tail -3 $O/t_step.log
# 2. kernels in isolation
python tools/attn_bench.py > $O/attn_bench.log 2>&1; cat $O/attn_bench.log
LT_ATTN_BWD_F2=0 python tools/attn_bench.py > $O/attn_bench_f2off.log 2>&1; grep local $O/attn_bench_f2off.log
# 3. in-step A/B of the new backward and of the stream schedules
python tools/ab_step.py LT_ATTN_BWD_F2 0 1 --steps 12 > $O/ab_f2.log 2>&1; tail -2 $O/ab_f2.log
python tools/ab_schedule.py --steps 10 five bwd2 two fwdmain one > $O/ab_schedule.log 2>&1; tail -5 $O/ab_schedule.log
# 4. bench lines: default, reference defaults (patch 14, drop-path 0.2)
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log | cut -c1-400
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --patch-size 14 > $O/bench_p14.log 2>&1; tail -1 $O/bench_p14.log | cut -c1-400
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --drop-path 0.2 > $O/bench_dp02.log 2>&1; tail -1 $O/bench_dp02.log | cut -c1-400
