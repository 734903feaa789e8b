#!/bin/bash
# round 6, GPU call 3: attn_pipe_kernel stream variants - distance between a VALU result and its first use (1 .. 4 instructions) x the
# 4-byte phase of the instruction stream (.p2align 3 + 0 / 1 s_nop).  Call 2: a lone wave ran the distance-1 stream at 8.8 cycles per
# instruction, and product / variant builds of one stream differed by 13 % (code placement).
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export O=gpurun_out/c3
mkdir -p $O
export ATTN_BENCH_SHAPES=0,1,2
echo "== product impl 4 / 5" >> $O/variants.txt
ATTN_BENCH_IMPLS=4,5 timeout 300 python tools/attn_bench.py >> $O/variants.txt 2>&1
for v in d1a0 d1a1 d2a0 d2a1 d3a0 d3a1 d4a0 d4a1; do
  echo "== variant $v" >> $O/variants.txt
  MMD_LIB=$PWD/mm-diffusion_amd/lib/variants/libmmd_$v.so ATTN_BENCH_IMPLS=5 timeout 300 python tools/attn_bench.py >> $O/variants.txt 2>&1
done
echo "== product impl 4 / 5 again" >> $O/variants.txt
ATTN_BENCH_IMPLS=4,5 timeout 300 python tools/attn_bench.py >> $O/variants.txt 2>&1
for v in d2a0 d3a0 d3a1; do
  echo "== variant $v, one block per CU" >> $O/variants.txt
  MMD_ATTN_PIPE_LDSPAD=65536 MMD_LIB=$PWD/mm-diffusion_amd/lib/variants/libmmd_$v.so ATTN_BENCH_IMPLS=5 timeout 300 python tools/attn_bench.py >> $O/variants.txt 2>&1
done
grep -v amdgpu.ids $O/variants.txt | sed 's/| dma-exact.*//' > $O/variants_clean.txt
cat $O/variants_clean.txt
