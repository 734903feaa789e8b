#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c24
mkdir -p $O
export MMD_POOL_NOREUSE=1
{ MMD_GEMM_STRIP=0 timeout 400 python tools/oob_check.py mid
  timeout 400 python tools/oob_check.py mid; } 2>&1 | grep -v amdgpu > $O/oob.txt
cut -c1-420 $O/oob.txt
