#!/bin/bash
R=$GRAFT_REPO_ROOT
T=${1:-r05j}
O=$R/gpurun_out/$T
mkdir -p $O
cd $R
export TMPDIR=/tmp
LT_GRAPH_BWD=1 timeout 120 python -X faulthandler tools/graph_probe.py 2>&1 | grep -E "^step|^OK|Error|error|Segm|File \"/root" | head -8
timeout 600 python -m pytest tests/test_gpu_step.py -q -x -k "hip_graph_replay" > $O/t_graph.log 2>&1; tail -5 $O/t_graph.log | cut -c1-250
LT_GRAPH_FWD=1 LT_GRAPH_BWD=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/bench_graph.log 2>&1; tail -1 $O/bench_graph.log | cut -c1-200
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/bench_nograph.log 2>&1; tail -1 $O/bench_nograph.log | cut -c1-200
LT_GRAPH_FWD=1 LT_GRAPH_BWD=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/bench_graph2.log 2>&1; tail -1 $O/bench_graph2.log | cut -c1-200
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/bench_nograph2.log 2>&1; tail -1 $O/bench_nograph2.log | cut -c1-200
