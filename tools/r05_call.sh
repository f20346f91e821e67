#!/bin/bash
R=$GRAFT_REPO_ROOT
T=${1:-r05n}
O=$R/gpurun_out/$T
mkdir -p $O
cd $R
export TMPDIR=/tmp
python tools/ab_schedule.py --steps 12 five tfirst > $O/ab_tfirst.log 2>&1; tail -2 $O/ab_tfirst.log | cut -c1-150
timeout 1500 python -m pytest tests/ -q -m gpu > $O/gpu_tests_full.log 2>&1; tail -2 $O/gpu_tests_full.log
