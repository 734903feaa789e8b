#!/bin/bash
# Round 3, GPU call 39: replay stress at the final build (more replays, full-size model, with a second process on the GPU)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c39
{ timeout 200 python tools/determinism_stress.py full 1500 noise
  timeout 100 python tools/determinism_stress.py mid 4000 noise
  MMD_GEMM_STRIP=0 timeout 100 python tools/determinism_stress.py mid 3000; } 2>&1 | grep -v amdgpu | tee gpurun_out/c39/stress.txt | cut -c1-200
