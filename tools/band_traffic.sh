#!/bin/bash
# L2 memory-side read traffic of the token GEMMs under different band widths of the tile order (LT_GEMM_BAND, read per call; "d" = the
# shipped default): rocprofv3 --pmc FETCH_SIZE over tools/gemm_bench.py (FETCH_SIZE x 2 per the gfx950 note, KiB -> MB).
#   bash tools/band_traffic.sh [kernel-name-pattern] [band ...]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/band; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
PAT=${1:-gemm256e}; shift
for b in ${@:-d 0 4}; do
  if [ "$b" = "d" ]; then unset LT_GEMM_BAND; else export LT_GEMM_BAND=$b; fi
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/f$b -o pmc -- python $R/tools/gemm_bench.py 11 tok-only > $O/run$b.log 2>&1
  F=$(find $O/f$b -name "*counter_collection.csv" | head -1)
  echo "== LT_GEMM_BAND=$b"; python $R/tools/pmc_kernels.py $PAT $F
  rm -rf $O/f$b
done
