#!/bin/bash
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['gemm_ms_per_step'], d['roofline']['traffic'])"; done
python bench.py --steps 6 --warmup 3 --no-cpu-baseline --drop-path 0.2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['gemm_time_method'][:60])"
