#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 600 python tools/ab_step.py two_bwd_chains 0 1 --attr --steps 20 2>&1 | tail -2
timeout 600 python tools/ab_step.py overlap_streams 0 1 --attr --steps 12 2>&1 | tail -2) | tee gpurun_out/r02y_sched_ab.log
