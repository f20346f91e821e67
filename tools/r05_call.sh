#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05r
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "center_column_sums or centering_without" > $O/t_ops.log 2>&1; tail -5 $O/t_ops.log | cut -c1-220
timeout 900 python -m pytest tests/test_gpu_step.py -q -x -k "bench_configuration or bitwise_reproducible or matches_reference_fixture or vits_width" > $O/t_step.log 2>&1; tail -3 $O/t_step.log | cut -c1-220
python tools/ab_step.py center_gemv 0 1 --attr --steps 16 > $O/ab_gemv.log 2>&1; tail -2 $O/ab_gemv.log
