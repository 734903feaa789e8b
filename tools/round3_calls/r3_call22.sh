#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c22
mkdir -p $O
{ MMD_POOL_NOREUSE=1 MMD_GEMM_STRIP=0 timeout 300 python tools/plan_neighbours.py mid 48 28 62; } 2>&1 | grep -v amdgpu > $O/nb.txt
cut -c1-330 $O/nb.txt
