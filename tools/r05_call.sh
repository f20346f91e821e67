#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05q
mkdir -p $O
cd $R
export TMPDIR=/tmp
python tools/ab_lib.py lightly-train_amd/lib/liblt_amd_f32nt.so --steps 24 > $O/ab_f32nt.log 2>&1; tail -3 $O/ab_f32nt.log | cut -c1-200
for lib in liblt_amd.so liblt_amd_f32nt.so; do echo "== $lib"; LT_AMD_LIB=$R/lightly-train_amd/lib/$lib python tools/gemm_bench.py 0 2>&1 | grep -E "head last fwd" ; done
