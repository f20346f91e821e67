#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05aa
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_step.py -q > $O/t_step.log 2>&1; tail -3 $O/t_step.log | cut -c1-200
python - <<'PY'
import json
for n in ("mid_koleo0.1", "vits_koleo0.1", "mid_koleo0.0", "vits_koleo0.0"):
    d = json.load(open(f"gpurun_out/trajectory_{n}.json")); print(n, d["hip_vs_reference_fp32"]["loss"])
PY
python tools/ab_schedule.py five --env LT_DXN_FULL_ZERO=1,0 --steps 20 2>&1 | grep -v amdgpu | tail -2
python bench.py --steps 20 --warmup 5 > $O/bench.log 2>&1; tail -1 $O/bench.log | cut -c1-200
