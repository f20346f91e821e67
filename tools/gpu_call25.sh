#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_step.py -m gpu -q -x -k "fixture or gradients_match or vitb_batch24 or trajectory" 2>&1 | grep -E "^E  |^FAILED|passed|failed" | cut -c1-300 | tail -6
timeout 600 python tools/ab_step.py s_head.pad_wgrad_rows 0 1 --attr --steps 25 2>&1 | tail -2 | tee gpurun_out/r02z_head_wgrad_ab.log
