#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05z
python tools/ab_schedule.py five --env LT_HACK_EARLY_LOCAL=0,1 --steps 20 2>&1 | grep -v amdgpu | tee gpurun_out/r05z/hack_early_local.log | tail -6
