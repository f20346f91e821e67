#!/bin/bash
# Scratch call script of round 4 (rewritten per gpurun call).  Last content: a same-process A/B of one per-call switch.
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/r04_call.sh <tag> <SWITCH> <v1> <v2> ...'
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r04x}
shift
mkdir -p $O
cd $R
export TMPDIR=/tmp
python tools/ab_step.py "$@" --steps 12 > $O/ab.log 2>&1; tail -4 $O/ab.log
