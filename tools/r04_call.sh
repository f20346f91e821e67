#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r04ao}
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_step.py -x -q -m gpu -k "fixture or bitwise or bench_conf or joint" > $O/step_tests.log 2>&1; tail -2 $O/step_tests.log
python tools/ab_step.py head_side 0 1 --attr --steps 15 > $O/ab_head_side.log 2>&1; tail -2 $O/ab_head_side.log
python tools/ab_step.py head_side 0 1 --attr --steps 15 > $O/ab_head_side2.log 2>&1; tail -2 $O/ab_head_side2.log
