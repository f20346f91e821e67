cd $GRAFT_REPO_ROOT; O=gpurun_out/r06m; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.log
python tools/outlier_probe.py --steps 300 2>&1 | grep -v amdgpu | tail -4 | tee $O/outlier_300.log
timeout 1700 python -m pytest tests/ -q -m gpu > $O/gpu_tests_full.log 2>&1; tail -3 $O/gpu_tests_full.log | tee $O/gpu_tests_tail.log
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_default_full.log; cut -c1-400 $O/bench_default_full.log
