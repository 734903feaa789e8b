#!/bin/bash
# gpurun --timeout 400 -- 'bash tools/gemm_timeline.sh'   : debug build -> per-block timelines of every microbench shape -> product build
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/timeline
MMD_EXTRA_CXXFLAGS=-DGEMM_TIMELINE python mm-diffusion_amd/build.py --force > gpurun_out/timeline/build.log 2>&1
timeout 250 python tools/gemm_timeline.py all gpurun_out/timeline/timeline.json > gpurun_out/timeline/timeline.txt 2>&1
python mm-diffusion_amd/build.py --force >> gpurun_out/timeline/build.log 2>&1
cat gpurun_out/timeline/timeline.txt
