#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
for g in 256 512 1024 256 512 1024; do echo -n "LT_LN_BWD_GRID=$g "; LT_LN_BWD_GRID=$g timeout 300 python bench.py --model vit_small --steps 20 --warmup 4 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*'; done | tee gpurun_out/r02w_vits_ln_grid.log

