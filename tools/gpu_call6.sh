#!/bin/bash
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -k "augment or multicrop or batchnorm or im2col or resnet or distillation" > gpurun_out/r02f_gpu_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r02f_gpu_tests.log
timeout 600 python bench.py --real-pipeline --no-cpu-baseline --no-roofline --steps 10 --warmup 3 > gpurun_out/r02f_bench_real_pipeline.log 2>&1
timeout 600 python bench.py --method distillationv3 --student resnet50 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r02f_bench_resnet50.log 2>&1
timeout 600 python bench.py --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/r02f_bench_default.log 2>&1
tail -12 gpurun_out/r02f_gpu_tests.log; tail -1 gpurun_out/r02f_bench_real_pipeline.log; tail -1 gpurun_out/r02f_bench_resnet50.log; tail -1 gpurun_out/r02f_bench_default.log
