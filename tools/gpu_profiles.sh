#!/bin/bash
# per-round profile collection on the GPU box (tag = first argument, e.g. r03): default bench line, rocprofv3 kernel stats (single-stream reference and shipped multi-stream),
# the three PMC passes (FETCH_SIZE | WRITE_SIZE | MFMA busy) over one default step, the per-workgroup GEMM timeline, cfg2 / cfg5 bench lines.
set -x
R=$GRAFT_REPO_ROOT
T=${1:-r02L}
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
for d in single multi; do python tools/rocprof_summary.py $(find $O/ks_$d -name "*.db" | head -1) 32 > $O/kernel_stats_$d.md 2>&1; done
python tools/step_timeline.py $(find $O/ks_multi -name "*.db" | head -1) 30 > $O/step_timeline.txt 2>&1
F=$(find $O/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1); W=$(find $O/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1); M=$(find $O/pmc_MFMA -name "*counter_collection.csv" | head -1)
python tools/pmc_step_traffic.py $F $W profiles/${T}_pmc_step_report.md > $O/gemm_traffic.txt 2>&1
python tools/pmc_step_report.py $F $W $M $FL > $O/pmc_step_report.md 2>&1
python tools/gemm_timeline.py > $O/gemm_timeline.txt 2>&1
$B --steps 10 --warmup 3 --model vit_small > $O/bench_vits.log 2>&1
(cd /tmp && rocprofv3 --kernel-trace --stats -d $O/ks_vits -o ks -- $B --steps 3 --warmup 1 --single-stream --model vit_small > $O/bench_vits_single.log 2>&1)
python tools/rocprof_summary.py $(find $O/ks_vits -name "*.db" | head -1) 32 > $O/kernel_stats_vits.md 2>&1
python tools/run_cfg5.py 32 > $O/cfg5.log 2>&1
$B --steps 10 --warmup 3 --method distillationv3 --student resnet50 > $O/bench_cfg4_resnet50.log 2>&1
# the N > 1 code path on this 1-GPU box: two ranks folded onto cuda:0 over gloo (RCCL refuses two ranks per device); the line's `comm`
# object carries the exposed all-reduce time per step -- a baseline to read the first real multi-GPU run against, not a scaling number
LT_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --batch 32 > $O/bench_gloo2.log 2>&1
timeout 1500 python -m pytest tests/ -q -m gpu > $O/gpu_tests_full.log 2>&1; grep -E "passed|failed" $O/gpu_tests_full.log | tail -1 > $O/gpu_tests_tail.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_default_b.log 2>&1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_default_c.log 2>&1
rm -rf $O/ks_* $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_MFMA
ls -la $O; cat $O/gpu_tests_tail.log; tail -1 $O/bench_default_b.log | cut -c1-200; tail -1 $O/bench_default_c.log | cut -c1-200; tail -3 $O/gemm_traffic.txt; tail -4 $O/pmc_step_report.md; tail -1 $O/bench_vits.log; tail -2 $O/cfg5.log; tail -1 $O/bench_default_full.log
