#!/bin/bash
# round-2 GPU call 2: new tests (conv ops, ResNet engine, ResNet-student distillation, ViT-B batch-24 step, resume, wrapper) + resnet50 bench
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --durations=10 -k "im2col or batchnorm or maxpool or token_mean or resnet or distillation or vitb_batch24 or koleo or resume or wrapper" > gpurun_out/r02c_gpu_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r02c_gpu_tests.log
cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r02c_prof_resnet50 -o r02c -- python $GRAFT_REPO_ROOT/bench.py --method distillationv3 --student resnet50 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/gpurun_out/r02c_bench_resnet50.log 2>&1; cd $GRAFT_REPO_ROOT; ls gpurun_out/r02c_prof_resnet50 | head; python tools/rocprof_summary.py $(find gpurun_out/r02c_prof_resnet50 -name "*.db" | head -1) 30 > gpurun_out/r02c_kernel_stats_resnet50.md 2>&1
tail -30 gpurun_out/r02c_gpu_tests.log; tail -3 gpurun_out/r02c_bench_resnet50.log
