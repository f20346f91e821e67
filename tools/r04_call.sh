#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r04l}
mkdir -p $O
cd $R
export TMPDIR=/tmp
python tools/ab_step.py LT_GEMM_1P 0 1 3 --steps 12 > $O/ab_gemm_1p.log 2>&1; tail -3 $O/ab_gemm_1p.log
python tools/ab_step.py LT_GEMM_1W 0 1 2 --steps 12 > $O/ab_gemm_1w.log 2>&1; tail -3 $O/ab_gemm_1w.log
