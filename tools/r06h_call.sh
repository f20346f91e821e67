cd $GRAFT_REPO_ROOT
for t in 1 0 1 0; do
  echo "== LT_GEMM_TAIL128=$t"
  LT_GEMM_TAIL128=$t timeout 600 python -m pytest tests/test_gpu_step.py -q -x -k "bench_configuration_step_matches" 2>&1 | grep -E "passed|failed|checks off" | cut -c1-300
done
