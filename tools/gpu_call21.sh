#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python tools/ab_step.py LT_GEMM_1W 0 1 2 --steps 20 2>&1 | tail -4 | tee gpurun_out/r02u_gemm1w_step_ab.log
