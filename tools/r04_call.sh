#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r04ak}
mkdir -p $O
cd $R
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
python bench.py > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log | cut -c1-260
