#!/bin/bash
# Rebuilds lightly-train_amd/lib/liblt_amd_1p.so: the CURRENT objects with gemm.hip replaced by the tree's last version that still carried the persistent
# 192 x 256 GEMM with the epilogue under the next tile's K-loop (gemm_p.hip, deleted in db361e8; gemm.hip has not changed since, so the variant's four-phase
# kernel is the shipped one).  `LT_GEMM_1P` (read per call: 1 = N >= 1024, 2 = narrower, 3 = all eligible forward / dgrad GEMMs) selects the persistent kernel.
#   bash tools/build_1p_variant.sh && LT_AMD_LIB=$PWD/lightly-train_amd/lib/liblt_amd_1p.so python tools/ab_schedule.py --env LT_GEMM_1P=0,1,3 five two one
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
W=${TMPDIR:-/tmp}/lt_1p_variant
mkdir -p $W/obj
for f in gemm.hip gemm_p.hip gemm_args.h; do git -C $R show db361e8^:lightly-train_amd/csrc/$f > $W/$f; done
python $R/__graft_entry__.py > /dev/null
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -I$W -I$R/lightly-train_amd/csrc -I$R/include -x hip"
hipcc $F -c $W/gemm.hip -o $W/obj/gemm_old.o
hipcc $F -c $W/gemm_p.hip -o $W/obj/gemm_p.o
OBJS=$(ls $R/lightly-train_amd/lib/obj/*.o | grep -v "gemm.hip.o\|gemm_timing.o\|gemm_f32nt.o")
hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $W/obj/gemm_old.o $W/obj/gemm_p.o -o $R/lightly-train_amd/lib/liblt_amd_1p.so
echo $R/lightly-train_amd/lib/liblt_amd_1p.so
