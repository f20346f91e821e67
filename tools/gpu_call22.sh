#!/bin/bash
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r02v_gpu_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r02v_gpu_tests.log
grep -E "^E  |^FAILED|passed|failed|rc=" gpurun_out/r02v_gpu_tests.log | cut -c1-300 | tail -12
bash tools/gpu_profiles.sh > gpurun_out/r02v_profiles.log 2>&1
tail -3 gpurun_out/r02v/gemm_traffic.txt | cut -c1-400
tail -1 gpurun_out/r02v/bench_default_full.log | cut -c1-300
