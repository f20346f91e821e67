#!/bin/bash
# L2 memory-side read traffic of the token GEMMs under the plain and the banded tile order of the four-phase kernel (LT_GEMM_BAND): rocprofv3 --pmc FETCH_SIZE over tools/gemm_bench.py (FETCH_SIZE x 2 per the gfx950 note, KiB -> MB).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/band; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
for b in 0 4; do
  LT_GEMM_BAND=$b rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/f$b -o pmc -- python $R/tools/gemm_bench.py 8 tok-only > $O/run$b.log 2>&1
  F=$(find $O/f$b -name "*counter_collection.csv" | head -1)
  echo "== LT_GEMM_BAND=$b"; python $R/tools/pmc_kernels.py gemm256q $F
done
rm -rf $O/f0 $O/f4
