#!/bin/bash
# round 6, GPU call 8 (timing-only ablations of attn_pipe_kernel's loop top): no K / V tile DMA in the loop, no barrier in the loop, neither.
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export O=gpurun_out/c8
mkdir -p $O
export ATTN_BENCH_SHAPES=0,1,2 ATTN_BENCH_IMPLS=5
for rep in 1 2; do
echo "== product (rep $rep)" >> $O/top.txt
timeout 300 python tools/attn_bench.py >> $O/top.txt 2>&1
for v in nodma nobar nodmanobar mfmaonly; do
  echo "== variant $v (rep $rep)" >> $O/top.txt
  MMD_LIB=$PWD/mm-diffusion_amd/lib/variants/libmmd_$v.so timeout 300 python tools/attn_bench.py >> $O/top.txt 2>&1
done
done
grep -v amdgpu.ids $O/top.txt | sed 's/| dma-exact.*//' > $O/top_clean.txt
cat $O/top_clean.txt
