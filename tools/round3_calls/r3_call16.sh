#!/bin/bash
# Round 3, GPU call 16: which switch removes the 1-in-300 replay difference (mid config, 1500 replays per setting)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c16
mkdir -p $O
N=1500
{ timeout 100 python tools/determinism_stress.py mid $N
  for kv in MMD_GN_EPILOGUE=0 MMD_GN_SMALL=0 MMD_ATTN_DMA=0 MMD_GEMM_DESC=0 MMD_HALO_GN=0 MMD_HALO16=0 MMD_GEMM_HALO=0 MMD_GEMM_RING=0 MMD_GEMM_STRIP=0; do
    env $kv timeout 100 python tools/determinism_stress.py mid $N
  done; } 2>&1 | grep -v amdgpu > $O/stress.txt
cut -c1-300 $O/stress.txt
