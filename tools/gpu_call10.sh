#!/bin/bash
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "attention" > gpurun_out/r02j_attn_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r02j_attn_tests.log
grep -E "^E  |^FAILED|passed|failed" gpurun_out/r02j_attn_tests.log | cut -c1-300 | tail -20
(echo "LT_ATTN_BWD=1"; LT_ATTN_BWD=1 timeout 120 python tools/attn_bench.py global 2>&1 | tail -1
echo "LT_ATTN_BWD=2"; timeout 120 python tools/attn_bench.py global 2>&1 | tail -1
for h in 1 4; do echo "LT_ATTN_BWD=2 HPB=$h"; LT_ATTN_BWD_HPB=$h timeout 120 python tools/attn_bench.py global 2>&1 | tail -1; done) | tee gpurun_out/r02j_attn_bench.log
