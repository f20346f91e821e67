#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r04al}
mkdir -p $O
cd $R
export TMPDIR=/tmp
for i in 1 2 3 4 5; do python bench.py --no-cpu-baseline --no-roofline > $O/bench_$i.log 2>&1; tail -1 $O/bench_$i.log | cut -c150-175; done
