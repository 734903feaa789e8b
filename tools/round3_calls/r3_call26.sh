#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c26
mkdir -p $O
export MMD_POOL_NOREUSE=1 MMD_GEMM_STRIP=0
V=mm-diffusion_amd/lib/variants
{ for n in gns1 gns2 gns3; do echo "## $n"; MMD_LIB=$V/libmmd_$n.so timeout 200 python tools/determinism_mini.py mid 41 50 40 200 | head -3; done
  echo "## product"; timeout 200 python tools/determinism_mini.py mid 41 50 40 200 | head -3; } 2>&1 | grep -v amdgpu > $O/mini.txt
cut -c1-400 $O/mini.txt
