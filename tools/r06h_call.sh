cd $GRAFT_REPO_ROOT; O=gpurun_out/r06m; mkdir -p $O
P='import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["config"].get("timed_region"))'
for i in 1 2 3 4 5 6; do
python bench.py --no-cpu-baseline --no-roofline --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "$P"
done | tee $O/bench_repeat.log
rocm-smi --showclocks --showpower --showtemp 2>/dev/null | head -30 | tee $O/smi.log
