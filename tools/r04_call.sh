#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r04ai}
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "layerscale" > $O/ls_tests.log 2>&1; tail -5 $O/ls_tests.log
