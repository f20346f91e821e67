#!/bin/bash
R=$GRAFT_REPO_ROOT
T=${1:-r05o}
O=$R/gpurun_out/$T
mkdir -p $O
cd $R
export TMPDIR=/tmp
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --drop-path 0.2 > $O/bench_dp02.log 2>&1; tail -1 $O/bench_dp02.log | cut -c1-200
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --drop-path 0.2 > $O/bench_dp02_b.log 2>&1; tail -1 $O/bench_dp02_b.log | cut -c1-200
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log | cut -c1-200
