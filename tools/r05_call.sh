#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05af
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -q -m gpu > $O/gpu_tests.log 2>&1; grep -a "passed\|failed" $O/gpu_tests.log | tail -2
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.log 2>&1; tail -1 $O/bench.log | cut -c1-200
