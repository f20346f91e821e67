cd $GRAFT_REPO_ROOT; O=gpurun_out/r06j; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_step.py -q -x -k "replay_of_the_static or bitwise_reproducible or binding_mixin or bench_configuration" 2>&1 | tail -3 | tee $O/tests.log
