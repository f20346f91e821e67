#!/bin/bash
# Round-6 profile collection on the GPU box (tag = first argument): the default bench line (roofline + thread-swept CPU baseline), rocprofv3 kernel
# stats (single-stream reference and the shipped five-stream schedule), the three PMC passes (FETCH_SIZE | WRITE_SIZE | MFMA busy) over one default
# step, the step timeline, the other BASELINE configurations WITH their roofline blocks (cfg2 ViT-S, cfg4 resnet50 student, cfg5 ViT-L/14 518^2),
# the reference's default geometry / regime (patch 14, SwiGLU, drop-path 0.2), the 2-rank gloo line, the GPU test tail.
set -x
R=$GRAFT_REPO_ROOT
T=${1:-r06}
O=$R/gpurun_out/$T
mkdir -p $O
export TMPDIR=/tmp
cd $R
python bench.py --steps 20 --warmup 5 > $O/bench_default_full.log 2>&1
FL=$(python -c "import json,sys; print(int(json.loads(open('$O/bench_default_full.log').read().strip().splitlines()[-1])['roofline']['gemm_flops_per_step']))")
cd /tmp
B="python $R/bench.py --no-cpu-baseline --no-roofline"
rocprofv3 --kernel-trace --stats -d $O/ks_single -o ks -- $B --steps 3 --warmup 1 --single-stream > $O/bench_single.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/ks_multi -o ks -- $B --steps 3 --warmup 1 > $O/bench_multi.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -o pmc -- $B --steps 1 --warmup 1 --single-stream > $O/pmc_$c.log 2>&1
done
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_MFMA -o pmc -- $B --steps 1 --warmup 1 --single-stream > $O/pmc_MFMA.log 2>&1
cd $R
for d in single multi; do python tools/rocprof_summary.py $(find $O/ks_$d -name "*.db" | head -1) 36 > $O/kernel_stats_$d.md 2>&1; done
python tools/step_timeline.py $(find $O/ks_multi -name "*.db" | head -1) 30 > $O/step_timeline.txt 2>&1
F=$(find $O/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1); W=$(find $O/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1); M=$(find $O/pmc_MFMA -name "*counter_collection.csv" | head -1)
python tools/pmc_step_traffic.py $F $W profiles/${T}_pmc_step_report.md > $O/gemm_traffic.txt 2>&1
python tools/pmc_step_report.py $F $W $M $FL > $O/pmc_step_report.md 2>&1
# the other BASELINE configurations, each with its roofline block
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --model vit_small > $O/bench_cfg2_vits.log 2>&1
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --method distillationv3 --student resnet50 > $O/bench_cfg4_resnet50.log 2>&1
python bench.py --steps 4 --warmup 2 --no-cpu-baseline --model vit_large --patch-size 14 --global-size 518 --ffn swiglufused --batch 32 > $O/bench_cfg5_vitl14_518.log 2>&1
# the reference's default geometry and regime
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --patch-size 14 > $O/bench_p14.log 2>&1
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --patch-size 14 --ffn swiglufused > $O/bench_p14_swiglu.log 2>&1
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --drop-path 0.2 > $O/bench_dp02.log 2>&1
for cfg in "dp02:--drop-path 0.2" "p14:--patch-size 14"; do
  n=${cfg%%:*}; f=${cfg#*:}
  (cd /tmp && rocprofv3 --kernel-trace --stats -d $O/ks_$n -o ks -- $B --steps 3 --warmup 1 --single-stream $f > $O/bench_${n}_single.log 2>&1)
  python tools/rocprof_summary.py $(find $O/ks_$n -name "*.db" | head -1) 36 > $O/kernel_stats_$n.md 2>&1
done
LT_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --batch 32 > $O/bench_gloo2.log 2>&1
timeout 1500 python -m pytest tests/ -q -m gpu > $O/gpu_tests_full.log 2>&1; grep -E "passed|failed" $O/gpu_tests_full.log | tail -1 > $O/gpu_tests_tail.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_default_b.log 2>&1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_default_c.log 2>&1
python tools/copybuffer_census.py $(find $O/ks_single -name "*.db" | head -1) > $O/copybuffer_census.log 2>&1
python tools/host_overhead.py vit_base > $O/host_overhead_plans.log 2>&1
LT_PLAN_FWD=0 LT_PLAN_BWD=0 python tools/host_overhead.py vit_base > $O/host_overhead_eager.log 2>&1
python tools/gemm_e_probe.py --kernels 8,11t0,11 --lib 1 > $O/gemm_e_probe.log 2>&1
python tools/attn_bench.py > $O/attn_bench.log 2>&1
rm -rf $O/ks_* $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_MFMA
set +x
ls $O; cat $O/gpu_tests_tail.log
for f in bench_default_full bench_default_b bench_default_c bench_cfg2_vits bench_cfg4_resnet50 bench_cfg5_vitl14_518 bench_p14 bench_p14_swiglu bench_dp02 bench_gloo2; do echo "$f: $(tail -1 $O/$f.log | cut -c1-160)"; done
tail -3 $O/gemm_traffic.txt | cut -c1-400; tail -4 $O/pmc_step_report.md
