#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c25
mkdir -p $O
export MMD_POOL_NOREUSE=1 MMD_GEMM_STRIP=0
{ timeout 200 python tools/determinism_mini.py mid 41 "" 40 200
  timeout 200 python tools/determinism_mini.py mid 41 50 40 200
  timeout 200 python tools/determinism_mini.py mid 41 47 40 200
  timeout 200 python tools/determinism_mini.py mid 41 45,46 40 200
  timeout 200 python tools/determinism_mini.py mid 40,41,42 50 40 400; } 2>&1 | grep -v amdgpu > $O/mini.txt
cut -c1-700 $O/mini.txt
