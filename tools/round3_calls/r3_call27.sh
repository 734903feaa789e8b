#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c27
mkdir -p $O
export MMD_POOL_NOREUSE=1 MMD_GEMM_STRIP=0
{ for a in 14 28 47 56 58 60; do timeout 100 python tools/determinism_mini.py mid 41 $a 40 100 | grep "replays differ"; done
  for v in 37 52 40 42 43 44 39 35; do timeout 100 python tools/determinism_mini.py mid $v 50 40 100 | grep "replays differ"; done; } 2>&1 | grep -v amdgpu > $O/mini.txt
cut -c1-200 $O/mini.txt
