#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r04aj}
mkdir -p $O
cd $R
export TMPDIR=/tmp
bash tools/clock_probe.sh "default bench step (ViT-B/16, batch 128), 150 steps" python bench.py --no-cpu-baseline --no-roofline --steps 150 --warmup 5 > $O/clock_probe.log 2>&1
bash tools/clock_probe.sh "single-stream schedule" python bench.py --no-cpu-baseline --no-roofline --steps 150 --warmup 5 --single-stream >> $O/clock_probe.log 2>&1
/opt/rocm/bin/rocm-smi --showmaxpower 2>/dev/null | grep -i "power" >> $O/clock_probe.log
cat $O/clock_probe.log
