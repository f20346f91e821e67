#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r04an}
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ddp.py -x -q -m gpu -k "joint or overlapped_with_backward" > $O/ddp_tests.log 2>&1; tail -12 $O/ddp_tests.log
