#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r04a}
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops_contract.py tests/test_gpu_dinov31.py -x -q -m gpu > $O/tests_dinov31.log 2>&1
tail -12 $O/tests_dinov31.log
timeout 900 python -m pytest tests/test_gpu_step.py -x -q -m gpu -k "mid_size" > $O/tests_traj.log 2>&1
tail -3 $O/tests_traj.log
timeout 200 python tools/host_ahead_probe.py --steps 30 2>&1 | grep -v amdgpu.ids | tail -4 > $O/host_ahead_default.log; cat $O/host_ahead_default.log
HSA_KERNARG_POOL_SIZE=33554432 timeout 200 python tools/host_ahead_probe.py --steps 30 2>&1 | grep -v amdgpu.ids | tail -4 > $O/host_ahead_pool32m.log; cat $O/host_ahead_pool32m.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log | cut -c1-500
