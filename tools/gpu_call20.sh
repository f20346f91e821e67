#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
for i in 1 2 3 4 5 6; do timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "gemm_256_kernel and 9" 2>&1 | grep -E "^FAILED|passed|failed" | cut -c1-200 | tail -5; done
timeout 300 python tools/gemm_bench.py 8,9 tok-only 2>&1 | tail -12 | tee gpurun_out/r02t_gemm1w.log
