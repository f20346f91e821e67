#!/bin/bash
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 300 python tools/host_overhead.py vit_base 2>&1 | grep -E "host-only|mask sampling" 
timeout 300 python tools/host_overhead.py vit_small 2>&1 | grep -E "host-only|mask sampling|tottime|\{|py:" | head -30) | tee gpurun_out/r02n_host_overhead.log
