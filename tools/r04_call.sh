#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r04s}
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "attention" > $O/attn_tests.log 2>&1; tail -3 $O/attn_tests.log
for v in 1 2 1 2; do echo "LT_ATTN_FWD_B2=$v"; LT_ATTN_FWD_B2=$v python tools/attn_bench.py 2>&1 | grep -v amdgpu | head -8; done > $O/attn_bench.log 2>&1; cat $O/attn_bench.log
python tools/ab_step.py LT_ATTN_FWD_B2 1 2 --steps 12 > $O/ab_attn_fwd.log 2>&1; tail -2 $O/ab_attn_fwd.log
