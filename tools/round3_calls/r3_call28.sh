#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c28
mkdir -p $O
export MMD_POOL_NOREUSE=1 MMD_GEMM_STRIP=0
V=mm-diffusion_amd/lib/variants
{ for n in gns4 gns5; do echo "## $n"; MMD_LIB=$V/libmmd_$n.so timeout 200 python tools/determinism_mini.py mid 41 50 40 200 | grep "replays differ"; MMD_LIB=$V/libmmd_$n.so timeout 200 python tools/determinism_mini.py mid 41 14 40 200 | grep "replays differ"; done
  echo "## product"; timeout 200 python tools/determinism_mini.py mid 41 50 40 200 | grep "replays differ"; } 2>&1 | grep -v amdgpu > $O/mini.txt
cut -c1-300 $O/mini.txt
