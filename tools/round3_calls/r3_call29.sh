#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c29
mkdir -p $O
export MMD_POOL_NOREUSE=1 MMD_GEMM_STRIP=0
V=mm-diffusion_amd/lib/variants
{ for n in gns6; do echo "## $n (no packed fp32 in mmd_norm)"; for a in 50 14 28; do MMD_LIB=$V/libmmd_$n.so timeout 200 python tools/determinism_mini.py mid 41 $a 40 300 | grep "replays differ"; done; done
  echo "## product"; timeout 200 python tools/determinism_mini.py mid 41 14 40 200 | grep "replays differ"; } 2>&1 | grep -v amdgpu > $O/mini.txt
cut -c1-300 $O/mini.txt
