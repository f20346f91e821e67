#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
(/opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power|power" | tr -s ' ' | tr '\n' ';'; echo " (idle)"
tools/clock_probe.sh "gemm fwd GEMMs (gemm_bench tok-only, 60 iterations each)" python tools/gemm_probe_loop.py
tools/clock_probe.sh "training step (bench.py --steps 120)" python bench.py --steps 120 --warmup 3 --no-cpu-baseline --no-roofline
tools/clock_probe.sh "mfma only (tools/mfma_peak)" bash -c "/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_peak tools/mfma_peak.hip && for i in 1 2 3 4 5 6 7 8; do /tmp/mfma_peak; done"
python tools/host_overhead.py vit_base 2>&1 | grep -E "host-only") 2>&1 | tee gpurun_out/r02o_clock_probe.log
