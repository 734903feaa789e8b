#!/bin/bash
# Round 3, GPU call 46: the vendor GEMM on the conv layers' plain GEMM shapes (yardstick)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c46
timeout 200 python tools/vendor_gemm_bench.py 2>&1 | grep -v amdgpu | tee gpurun_out/c46/vendor_gemm.txt
