#!/bin/bash
# round-2 GPU call 1: full -m gpu suite, default bench, 100-step trajectories vs the reference fixture
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x --durations=15 > gpurun_out/r02a_gpu_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r02a_gpu_tests.log
timeout 600 python bench.py > gpurun_out/r02a_bench.log 2>&1
timeout 300 python tests/tools/trajectory.py --steps 100 --koleo 0.0 > gpurun_out/r02a_traj_k0.log 2>&1
timeout 300 python tests/tools/trajectory.py --steps 100 --koleo 0.1 > gpurun_out/r02a_traj_k01.log 2>&1
tail -5 gpurun_out/r02a_gpu_tests.log; tail -2 gpurun_out/r02a_bench.log; tail -1 gpurun_out/r02a_traj_k0.log; tail -1 gpurun_out/r02a_traj_k01.log
