#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r04a}
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/ -x -q -m gpu > $O/gpu_tests.log 2>&1
tail -6 $O/gpu_tests.log
cd /tmp
B="python $R/bench.py --no-cpu-baseline --no-roofline"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/ks_vits -o ks -- $B --steps 3 --warmup 1 --single-stream --model vit_small > $O/bench_vits_single.log 2>&1
cd $R
python tools/rocprof_summary.py $(find $O/ks_vits -name "*.db" | head -1) 36 > $O/kernel_stats_vits.md 2>&1
rm -rf $O/ks_vits
head -42 $O/kernel_stats_vits.md
$B --steps 10 --warmup 3 --model vit_small > $O/bench_vits.log 2>&1; tail -1 $O/bench_vits.log | cut -c1-300
timeout 200 python tools/host_overhead.py 2>&1 | grep -v amdgpu | head -30 > $O/host_overhead.log; cat $O/host_overhead.log
