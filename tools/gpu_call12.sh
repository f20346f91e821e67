#!/bin/bash
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python tools/ab_step.py LT_ATTN_BWD 1 2 --steps 30 2>&1 | tail -3 | tee gpurun_out/r02l_attn_step_ab.log
