#!/bin/bash
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python tools/resnet_debug.py > gpurun_out/r02d_resnet_debug.log 2>&1
timeout 900 python -m pytest tests -m gpu -q -k "token_mean or koleo_gradients or resume or wrapper" > gpurun_out/r02d_gpu_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r02d_gpu_tests.log
cat gpurun_out/r02d_resnet_debug.log; tail -15 gpurun_out/r02d_gpu_tests.log
