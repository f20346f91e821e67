#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05ac
mkdir -p $O
export TMPDIR=/tmp
cd $R
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log | cut -c1-200
python bench.py --steps 20 --warmup 5 > $O/bench_default_full.log 2>&1; tail -1 $O/bench_default_full.log | cut -c1-180
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_default_b.log 2>&1; tail -1 $O/bench_default_b.log | cut -c1-180
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --drop-path 0.2 > $O/bench_dp02.log 2>&1; tail -1 $O/bench_dp02.log | cut -c1-180
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --patch-size 14 > $O/bench_p14.log 2>&1; tail -1 $O/bench_p14.log | cut -c1-180
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/ks_multi -o ks -- python $R/bench.py --no-cpu-baseline --no-roofline --steps 3 --warmup 1 > $O/bench_multi.log 2>&1
cd $R
python tools/rocprof_summary.py $(find $O/ks_multi -name "*.db" | head -1) 36 > $O/kernel_stats_multi.md 2>&1
python tools/step_timeline.py $(find $O/ks_multi -name "*.db" | head -1) 30 > $O/step_timeline.txt 2>&1
rm -rf $O/ks_multi
head -5 $O/step_timeline.txt | cut -c1-200
