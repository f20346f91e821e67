#!/bin/bash
R=$GRAFT_REPO_ROOT
T=${1:-r05g}
O=$R/gpurun_out/$T
mkdir -p $O
cd $R
export TMPDIR=/tmp
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --drop-path 0.2 > $O/bench_dp02.log 2>&1; tail -1 $O/bench_dp02.log | cut -c1-200
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log | cut -c1-200
python tools/host_profile.py --drop-path 0.2 --steps 12 > $O/host_profile_dp02.log 2>&1; grep -v "^$" $O/host_profile_dp02.log | head -10 | cut -c1-180
python tools/host_profile.py --steps 12 > $O/host_profile_default.log 2>&1; grep -v "^$" $O/host_profile_default.log | head -12 | cut -c1-180
timeout 1500 python -m pytest tests/ -q -m gpu > $O/gpu_tests_full.log 2>&1; tail -3 $O/gpu_tests_full.log
