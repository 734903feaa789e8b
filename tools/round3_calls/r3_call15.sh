#!/bin/bash
# Round 3, GPU call 15: replay stress for run-to-run differences, quiet and with a second process on the GPU
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c15
mkdir -p $O
{ timeout 120 python tools/determinism_stress.py mid 300
  timeout 120 python tools/determinism_stress.py mid 300 noise
  timeout 200 python tools/determinism_stress.py full 150 noise
  MMD_GN_TAIL=all timeout 120 python tools/determinism_stress.py mid 300 noise; } 2>&1 | grep -v amdgpu > $O/stress.txt
cut -c1-400 $O/stress.txt
