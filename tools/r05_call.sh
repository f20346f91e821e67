#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05s
mkdir -p $O
cd $R
export TMPDIR=/tmp
LT_AMD_LIB=$R/lightly-train_amd/lib/liblt_amd_1p.so python tools/ab_schedule.py --steps 10 --env LT_GEMM_1P=0,1,3 five two one > $O/ab_sched_1p.log 2>&1; tail -10 $O/ab_sched_1p.log | cut -c1-160
