#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05ag
mkdir -p $O
cd $R
LT_AMD_LIB=$R/lightly-train_amd/lib/liblt_amd_epipre.so timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "gemm" > $O/t_gemm.log 2>&1; tail -1 $O/t_gemm.log | cut -c1-200
for lib in liblt_amd_epinopre.so liblt_amd_epipre.so liblt_amd_epinopre.so liblt_amd_epipre.so; do LT_AMD_LIB=$R/lightly-train_amd/lib/$lib python tools/gemm_resid_probe.py 2>&1 | grep -v amdgpu; done | tee $O/resid_probe.log
LT_AMD_LIB=$R/lightly-train_amd/lib/liblt_amd_epinopre.so python tools/ab_lib.py lightly-train_amd/lib/liblt_amd_epipre.so --steps 16 > $O/ab_epipre.log 2>&1; tail -3 $O/ab_epipre.log | cut -c1-200
