#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05u
mkdir -p $O
cd $R
LT_AMD_LIB=$R/lightly-train_amd/lib/liblt_amd_dmasched.so timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "gemm" > $O/t_gemm.log 2>&1; tail -2 $O/t_gemm.log | cut -c1-200
for lib in liblt_amd.so liblt_amd_dmasched.so; do LT_AMD_LIB=$R/lightly-train_amd/lib/$lib python tools/gemm_kloop_probe.py 2>&1 | grep -v amdgpu; done
python tools/ab_lib.py lightly-train_amd/lib/liblt_amd_dmasched.so --steps 16 > $O/ab_dmasched.log 2>&1; tail -3 $O/ab_dmasched.log | cut -c1-200
