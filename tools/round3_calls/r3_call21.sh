#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c21
mkdir -p $O
{ MMD_POOL_NOREUSE=1 MMD_GEMM_STRIP=0 timeout 300 python tools/determinism_graph.py mid 2500; } 2>&1 | grep -v amdgpu > $O/graph_b.txt
cut -c1-260 $O/graph_b.txt | grep -v "^mid:" | head -150
