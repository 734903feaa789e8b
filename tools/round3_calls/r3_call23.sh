#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c23
mkdir -p $O
export MMD_POOL_NOREUSE=1 MMD_GEMM_STRIP=0
{ timeout 200 python tools/determinism_mini.py mid 40,41,42 45,46,47,48,49,50,54,55,56 40 300
  timeout 200 python tools/determinism_mini.py mid 40,41,42 "" 40 300
  timeout 200 python tools/determinism_mini.py mid 41 45,46,47,48,49,50,54,55,56 40 300
  timeout 200 python tools/determinism_mini.py mid 40,41,42 47 40 300
  timeout 200 python tools/determinism_mini.py mid 40,41,42 48 40 300
  timeout 200 python tools/determinism_mini.py mid 40,41,42 50 40 300
  timeout 200 python tools/determinism_mini.py mid 40,41,42 45,46 40 300
  timeout 200 python tools/determinism_mini.py mid 40,41,42 49,54,55 40 300; } 2>&1 | grep -v amdgpu > $O/mini.txt
cut -c1-300 $O/mini.txt
