#!/bin/bash
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r02p_gpu_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r02p_gpu_tests.log
grep -E "^E  |^FAILED|passed|failed|rc=" gpurun_out/r02p_gpu_tests.log | cut -c1-300 | tail -12
timeout 600 python bench.py > gpurun_out/r02p_bench_default.json.log 2>&1; tail -1 gpurun_out/r02p_bench_default.json.log | cut -c1-1500
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
