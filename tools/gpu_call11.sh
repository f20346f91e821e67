#!/bin/bash
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02k
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for v in 1 2; do
  LT_ATTN_BWD=$v rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/p1_$v -o pmc -- python $R/tools/attn_bench.py global > $O/p1_$v.log 2>&1
  LT_ATTN_BWD=$v rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/p2_$v -o pmc -- python $R/tools/attn_bench.py global > $O/p2_$v.log 2>&1
  LT_ATTN_BWD=$v rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS --output-format csv -d $O/p3_$v -o pmc -- python $R/tools/attn_bench.py global > $O/p3_$v.log 2>&1
done
cd $R
for v in 1 2; do
  echo "=== LT_ATTN_BWD=$v"
  python tools/pmc_kernels.py attn_bwd $(find $O/p1_$v $O/p2_$v $O/p3_$v -name "*counter_collection.csv")
done > $O/attn_bwd_pmc.txt 2>&1
tail -3 $O/p1_2.log $O/p2_2.log $O/p3_2.log
rm -rf $O/p1_* $O/p2_* $O/p3_*
cat $O/attn_bwd_pmc.txt
