#!/bin/bash
cd $GRAFT_REPO_ROOT
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | cut -c80-200
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --drop-path 0.2 2>/dev/null | tail -1 | cut -c80-200
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | cut -c80-200
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --drop-path 0.2 2>/dev/null | tail -1 | cut -c80-200
