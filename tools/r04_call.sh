#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r04z}
mkdir -p $O
cd $R
export TMPDIR=/tmp
python tools/ab_lib.py lightly-train_amd/lib/liblt_amd_old.so --steps 15 > $O/ab_lib_old.log 2>&1; tail -3 $O/ab_lib_old.log
python tools/ab_step.py joint_wgrad 0 1 --attr --steps 10 > $O/ab_joint.log 2>&1; tail -2 $O/ab_joint.log
