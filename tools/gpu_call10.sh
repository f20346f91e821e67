#!/bin/bash
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "attention" > gpurun_out/r02j_attn_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r02j_attn_tests.log
grep -E "^E  |^FAILED|passed|failed" gpurun_out/r02j_attn_tests.log | cut -c1-300 | tail -20
for v in 1 2; do echo "LT_ATTN_BWD=$v"; LT_ATTN_BWD=$v timeout 120 python tools/attn_bench.py 2>&1 | tail -5; done | tee gpurun_out/r02j_attn_bench.log
