cd $GRAFT_REPO_ROOT; O=gpurun_out/r06h; mkdir -p $O
python tools/host_overhead.py vit_base > $O/host_plan_full.log 2>&1
python tools/host_overhead.py vit_base 0.2 > $O/host_dp02_full.log 2>&1
grep -E "host-only" $O/host_plan_full.log $O/host_dp02_full.log
