#!/bin/bash
# Round 3, GPU call 17: where in the plan does the 1-in-130 replay difference start
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c17
mkdir -p $O
{ timeout 100 python tools/determinism_trace.py mid seq 1500
  timeout 100 python tools/determinism_trace.py mid two 1500
  MMD_GEMM_STRIP=0 timeout 100 python tools/determinism_trace.py mid seq 600
  MMD_GEMM_STRIP=0 timeout 300 python tools/determinism_trace.py mid trace 150; } 2>&1 | grep -v amdgpu > $O/trace.txt
cut -c1-300 $O/trace.txt
