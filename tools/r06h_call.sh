cd $GRAFT_REPO_ROOT; O=gpurun_out/r06l; mkdir -p $O
python tools/plan_ab_probe.py vit_small 10 6 2>&1 | grep -v amdgpu | tee $O/plan_ab_vits.log
python tools/plan_ab_probe.py vit_base 8 5 2>&1 | grep -v amdgpu | tee $O/plan_ab_vitb.log
