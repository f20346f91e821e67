cd $GRAFT_REPO_ROOT; O=gpurun_out/r06m; mkdir -p $O
python tools/plan_ab_probe.py vit_small 24 6 2>&1 | grep -v amdgpu | tee $O/plan_ab_vits_long.log
python tools/plan_ab_probe.py vit_small 24 6 2>&1 | grep -v amdgpu | tee -a $O/plan_ab_vits_long.log
