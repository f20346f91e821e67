#!/bin/bash
# rocprofv3 --pmc passes over one GEMM shape for the four-phase (8) and static-address (11) kernels (tools/gemm_one.py); output: gpurun_out/<tag>_gemm_loop_pmc.log
# usage: tools/gemm_loop_pmc.sh <tag> [M N K tb]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r06}; M=${2:-4096}; N=${3:-4096}; K=${4:-8192}; TB=${5:-0}
OUT=$R/gpurun_out/${TAG}_gemm_loop_pmc.log
: > $OUT
for FK in 8 11; do
  for SET in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA"; do
    D=/tmp/pmc_${FK}_$RANDOM
    rocprofv3 --pmc $SET --kernel-trace -d $D -o p --output-format csv -- python $R/tools/gemm_one.py $M $N $K $TB $FK > /dev/null 2>&1
    echo "== $M $N $K tb=$TB kernel $FK | $SET" >> $OUT
    python $R/tools/pmc_kernels.py gemm256 $(find $D -name "*counter_collection.csv") >> $OUT 2>&1
  done
done
cat $OUT
