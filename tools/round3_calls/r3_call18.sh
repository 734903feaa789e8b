#!/bin/bash
# Round 3, GPU call 18: which buffers differ after a differing graph replay
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c18
mkdir -p $O
{ MMD_GEMM_STRIP=0 timeout 200 python tools/determinism_graph.py mid 400
  timeout 200 python tools/determinism_graph.py mid 1500; } 2>&1 | grep -v amdgpu > $O/graph.txt
cut -c1-260 $O/graph.txt | head -120
