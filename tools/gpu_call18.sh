#!/bin/bash
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "attention" 2>&1 | grep -E "^E  |^FAILED|passed|failed" | cut -c1-300 | tail -8
(python tools/attn_bench.py 2>&1 | tail -5; LT_AMD_LIB=$PWD/lightly-train_amd/lib/liblt_amd_variant.so python tools/attn_bench.py 2>&1 | tail -5
timeout 600 python tools/ab_lib.py lightly-train_amd/lib/liblt_amd_variant.so --steps 30 2>&1 | tail -3) | tee gpurun_out/r02r_attn_fwd_ab.log
