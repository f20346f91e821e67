#!/bin/bash
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -k "distill" > gpurun_out/r02g_gpu_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r02g_gpu_tests.log
grep -E "^E  |^FAILED|passed|failed" gpurun_out/r02g_gpu_tests.log | cut -c1-300 | tail -30
