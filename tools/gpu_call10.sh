#!/bin/bash
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 120 python tools/attn_bench.py ab LT_ATTN_BWD_HPB 1 2 3 4 6 12 2>&1 | tail -6
timeout 120 python tools/attn_bench.py ab LT_ATTN_BWD 1 2 2>&1 | tail -2) | tee gpurun_out/r02j_attn_ab.log
