#!/bin/bash
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --durations=8 > gpurun_out/r02e_gpu_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r02e_gpu_tests.log
tail -25 gpurun_out/r02e_gpu_tests.log
