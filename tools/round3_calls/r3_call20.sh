#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c20
mkdir -p $O
{ MMD_GEMM_STRIP=0 timeout 300 python tools/determinism_graph2.py mid 300; timeout 300 python tools/determinism_graph2.py mid 1000; } 2>&1 | grep -v amdgpu > $O/graph2.txt
cut -c1-200 $O/graph2.txt
