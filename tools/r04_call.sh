#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r04p}
mkdir -p $O
cd $R
export TMPDIR=/tmp
for i in 1 2 3; do
python tools/ab_step.py joint_wgrad 0 1 --attr --steps 10 > $O/ab_q4_$i.log 2>&1; tail -2 $O/ab_q4_$i.log
done
