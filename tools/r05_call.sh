#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_ops.py -q -x -k "row_owning" 2>&1 | tail -4 | cut -c1-200
