#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r04a}
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_ops_contract.py -x -q -m gpu > $O/tests_ops.log 2>&1
tail -3 $O/tests_ops.log
timeout 900 python -m pytest tests/test_gpu_step.py -x -q -m gpu -k "mid_size or trajectory or bench_configuration" > $O/tests_traj.log 2>&1
tail -3 $O/tests_traj.log
timeout 300 python tools/outlier_probe.py --steps 150 > $O/outlier_gc_on.log 2>&1; cat $O/outlier_gc_on.log | tail -25
timeout 300 python tools/outlier_probe.py --steps 150 --gc-off > $O/outlier_gc_off.log 2>&1; cat $O/outlier_gc_off.log | tail -25
timeout 300 python tools/ab_step.py fused_centering 0 1 --attr --steps 25 > $O/ab_fused_centering.log 2>&1; tail -2 $O/ab_fused_centering.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log | cut -c1-400
