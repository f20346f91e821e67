#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05ab
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/ks_single -o ks -- python $R/bench.py --no-cpu-baseline --no-roofline --steps 3 --warmup 1 --single-stream > $O/bench_single.log 2>&1
cd $R
python tools/rocprof_summary.py $(find $O/ks_single -name "*.db" | head -1) 90 > $O/kernel_stats_single_90.md 2>&1
rm -rf $O/ks_single
tail -60 $O/kernel_stats_single_90.md | cut -c1-150
