#!/bin/bash
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "attention" 2>&1 | grep -E "^E  |^FAILED|passed|failed" | cut -c1-300 | tail -8
(timeout 120 python tools/attn_bench.py ab LT_ATTN_BWD 1 2 2>&1 | tail -2; timeout 120 python tools/attn_bench.py 2>&1 | tail -5) | tee gpurun_out/r02j_attn_ab2.log
