#!/bin/bash
cd $GRAFT_REPO_ROOT
python tools/joint_gemm_probe.py 2>&1 | grep -v amdgpu.ids
