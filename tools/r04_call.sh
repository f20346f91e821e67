#!/bin/bash
# reduced re-collection at the final tree (no PMC passes: gemm.hip is unchanged since the full `tools/gpu_profiles.sh r04` call)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r04fin}
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 1500 python -m pytest tests/ -q -m gpu > $O/gpu_tests_full.log 2>&1; grep -E "passed|failed" $O/gpu_tests_full.log | tail -1 > $O/gpu_tests_tail.log
python bench.py --steps 20 --warmup 5 > $O/bench_default_full.log 2>&1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_default_b.log 2>&1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_default_c.log 2>&1
cd /tmp
B="python $R/bench.py --no-cpu-baseline --no-roofline"
rocprofv3 --kernel-trace --stats -d $O/ks_single -o ks -- $B --steps 3 --warmup 1 --single-stream > $O/bench_single.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/ks_multi -o ks -- $B --steps 3 --warmup 1 > $O/bench_multi.log 2>&1
cd $R
for d in single multi; do python tools/rocprof_summary.py $(find $O/ks_$d -name "*.db" | head -1) 32 > $O/kernel_stats_$d.md 2>&1; done
python tools/step_timeline.py $(find $O/ks_multi -name "*.db" | head -1) 30 > $O/step_timeline.txt 2>&1
$B --steps 10 --warmup 3 --model vit_small > $O/bench_vits.log 2>&1
$B --steps 10 --warmup 3 --method distillationv3 --student resnet50 > $O/bench_cfg4_resnet50.log 2>&1
rm -rf $O/ks_*
cat $O/gpu_tests_tail.log; for f in bench_default_full bench_default_b bench_default_c bench_vits bench_cfg4_resnet50; do tail -1 $O/$f.log | cut -c1-190; done; sed -n 3,4p $O/step_timeline.txt
