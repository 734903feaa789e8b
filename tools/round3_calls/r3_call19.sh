#!/bin/bash
# Round 3, GPU call 19: bisect the plan for the launch behind the differing graph replays
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c19
mkdir -p $O
{ MMD_GEMM_STRIP=0 timeout 300 python tools/determinism_bisect.py mid 80
  MMD_GEMM_STRIP=0 BISECT_SINGLE_STREAM=1 timeout 300 python tools/determinism_bisect.py mid 80; } 2>&1 | grep -v amdgpu > $O/bisect.txt
cut -c1-200 $O/bisect.txt
