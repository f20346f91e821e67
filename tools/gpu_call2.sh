#!/bin/bash
# round-2 GPU call 2: new tests (conv ops, ResNet engine, ResNet-student distillation, ViT-B batch-24 step, resume, wrapper) + resnet50 bench
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --durations=10 -k "im2col or batchnorm or maxpool or token_mean or resnet or distillation or vitb_batch24 or koleo or freeze or resume or wrapper or trajectory or attention" > gpurun_out/r02b_gpu_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r02b_gpu_tests.log
timeout 600 python bench.py --method distillationv3 --student resnet50 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r02b_bench_resnet50.log 2>&1
tail -30 gpurun_out/r02b_gpu_tests.log; tail -3 gpurun_out/r02b_bench_resnet50.log
