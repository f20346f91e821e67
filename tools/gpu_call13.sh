#!/bin/bash
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python tools/gemm_bench_vits.py 2>&1 | tail -12 | tee gpurun_out/r02m_gemm_vits.log
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "gemm" 2>&1 | grep -E "^E  |^FAILED|passed|failed" | cut -c1-300 | tail -8
