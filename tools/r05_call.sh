#!/bin/bash
cd $GRAFT_REPO_ROOT
python tools/host_overhead.py 2>&1 | grep "host-only"
LT_GRAPH_FWD=1 python tools/host_overhead.py 2>&1 | grep "host-only"
LT_GRAPH_FWD=1 LT_GRAPH_BWD=1 python tools/host_overhead.py 2>&1 | grep "host-only"
