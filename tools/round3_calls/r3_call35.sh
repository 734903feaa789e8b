#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/c35
mkdir -p $O
timeout 200 python tools/graph_node_cost.py 2>&1 | grep -v amdgpu | tee $O/node_cost.txt
