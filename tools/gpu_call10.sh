#!/bin/bash
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "attention" 2>&1 | grep -E "^E  |^FAILED|passed|failed" | cut -c1-300 | tail -12
(for v in 0 1; do echo "LT_ATTN_BWD_PACK=$v"; LT_ATTN_BWD_PACK=$v timeout 120 python tools/attn_bench.py 2>&1 | tail -5 | head -3; done) | tee gpurun_out/r02x_attn_pack.log
