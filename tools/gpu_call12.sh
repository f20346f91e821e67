#!/bin/bash
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python tools/ab_step.py LT_ATTN_BWD_HPB 2 4 6 12 --steps 20 2>&1 | tail -4 | tee gpurun_out/r02l_hpb_step_ab.log
