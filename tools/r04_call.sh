#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r04aa}
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_distill.py -x -q -m gpu -k "batchnorm or bn or resnet" > $O/bn_tests.log 2>&1; tail -2 $O/bn_tests.log
B="python $R/bench.py --no-cpu-baseline --no-roofline --method distillationv3 --student resnet50"
$B --steps 10 --warmup 3 > $O/bench_cfg4_a.log 2>&1; tail -1 $O/bench_cfg4_a.log | cut -c100-230
$B --steps 10 --warmup 3 > $O/bench_cfg4_b.log 2>&1; tail -1 $O/bench_cfg4_b.log | cut -c100-230
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/ks -o ks -- $B --steps 3 --warmup 1 --single-stream > $O/bench_cfg4_single.log 2>&1
cd $R
python tools/rocprof_summary.py $(find $O/ks -name "*.db" | head -1) 40 > $O/kernel_stats_cfg4.md 2>&1
rm -rf $O/ks
grep -E "bn_|total kernel" $O/kernel_stats_cfg4.md
