#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r04ah}
mkdir -p $O
cd $R
export TMPDIR=/tmp
python tools/ab_step.py LT_SHARED_COLS 0 1 --steps 15 > $O/ab_cols.log 2>&1; tail -2 $O/ab_cols.log
python tools/ab_step.py LT_GRAD_ZERO_SIDE 0 1 --steps 15 > $O/ab_zero.log 2>&1; tail -2 $O/ab_zero.log
python tools/ab_step.py LT_SHARED_COLS 0 1 --steps 15 > $O/ab_cols2.log 2>&1; tail -2 $O/ab_cols2.log
