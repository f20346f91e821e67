#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05y
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -q -m gpu > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log | cut -c1-200
python bench.py --steps 20 --warmup 5 > $O/bench.log 2>&1; tail -1 $O/bench.log | cut -c1-200
python bench.py --steps 20 --warmup 5 --drop-path 0.2 > $O/bench_dp02.log 2>&1; tail -1 $O/bench_dp02.log | cut -c1-200
