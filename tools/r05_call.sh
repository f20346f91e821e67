#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05ae
mkdir -p $O
cd $R
for i in 1 2 3; do timeout 600 python -m pytest tests/test_gpu_ddp.py -q > $O/ddp_$i.log 2>&1; grep -a "passed\|failed" $O/ddp_$i.log | tail -1; done
timeout 600 python -m pytest tests/test_gpu_ddp.py -q -x > $O/ddp_default.log 2>&1; grep -a "passed\|failed" $O/ddp_default.log | tail -1
