cd $GRAFT_REPO_ROOT; O=gpurun_out/r06i; mkdir -p $O
python tools/dp_stall_probe.py 0.2 2>&1 | grep -v amdgpu | tee $O/dp_stall.log | head -80
