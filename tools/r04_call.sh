#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r04ap}
mkdir -p $O
cd $R
export TMPDIR=/tmp
python tools/ab_step.py LT_GEMM_WGRAD_MAXK 0 8192 16384 --steps 12 > $O/ab_maxk.log 2>&1; tail -3 $O/ab_maxk.log
