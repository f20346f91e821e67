#!/bin/bash
R=$GRAFT_REPO_ROOT
T=${1:-r05l}
O=$R/gpurun_out/$T
mkdir -p $O
cd $R
export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-roofline --steps 20 --warmup 5"
for e in "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "DEBUG_HIP_FORCE_GRAPH_QUEUES=8" "DEBUG_HIP_FORCE_GRAPH_QUEUES=4" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 DEBUG_HIP_FORCE_GRAPH_QUEUES=8" "DEBUG_HIP_GRAPH_BATCH_SIZE=1" "X=1"; do
  echo "== $e: $(env $e LT_GRAPH_BWD=1 $B 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*')"
done
echo "== eager: $($B 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*')"
