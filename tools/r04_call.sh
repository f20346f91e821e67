#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r04a}
mkdir -p $O
cd $R
export TMPDIR=/tmp
run() { echo "== $*"; env "$@" timeout 200 python tools/host_ahead_probe.py --steps 30 2>&1 | grep -v amdgpu.ids | tail -4; }
{
run A=0
run HSA_KERNARG_POOL_SIZE=33554432
run HSA_KERNARG_POOL_SIZE=33554432 ROC_AQL_QUEUE_SIZE=65536
run ROC_SIGNAL_POOL_SIZE=4096
run HSA_KERNARG_POOL_SIZE=33554432 ROC_AQL_QUEUE_SIZE=65536 ROC_SIGNAL_POOL_SIZE=4096 GPU_MAX_HW_QUEUES=8
run HIP_FORCE_DEV_KERNARG=1
} > $O/host_ahead_env.log 2>&1
cat $O/host_ahead_env.log
