#!/bin/bash
# round-4 GPU call: op tests of the new kernels, in-step A/B of their switches, per-kernel profile, default bench line
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r04a}
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_ops_contract.py -x -q -m gpu > $O/tests_ops.log 2>&1
tail -3 $O/tests_ops.log
timeout 300 python tools/ab_step.py fused_centering 0 1 --attr --steps 25 > $O/ab_fused_centering.log 2>&1; tail -2 $O/ab_fused_centering.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log
cd /tmp
B="python $R/bench.py --no-cpu-baseline --no-roofline"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/ks_single -o ks -- $B --steps 3 --warmup 1 --single-stream > $O/bench_single.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $O/ks_multi -o ks -- $B --steps 3 --warmup 1 > $O/bench_multi.log 2>&1
cd $R
for d in single multi; do python tools/rocprof_summary.py $(find $O/ks_$d -name "*.db" | head -1) 40 > $O/kernel_stats_$d.md 2>&1; done
rm -rf $O/ks_single $O/ks_multi
head -45 $O/kernel_stats_single.md
