#!/bin/bash
# round 6, GPU call 4: is "product impl 5 = 114 us, every variant build = 99 us" a measurement-order artefact (impl 5 timed right behind impl 4)?
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export O=gpurun_out/c4
mkdir -p $O
export ATTN_BENCH_SHAPES=0,1,2
for order in 5 4 5,4 4,5 5,5 4,4; do
  echo "== product, impls $order" >> $O/order.txt
  ATTN_BENCH_IMPLS=$order timeout 300 python tools/attn_bench.py >> $O/order.txt 2>&1
done
for order in 5 4,5; do
  echo "== variant d3a0, impls $order" >> $O/order.txt
  MMD_LIB=$PWD/mm-diffusion_amd/lib/variants/libmmd_d3a0.so ATTN_BENCH_IMPLS=$order timeout 300 python tools/attn_bench.py >> $O/order.txt 2>&1
done
grep -v amdgpu.ids $O/order.txt | sed 's/| dma-exact.*//' > $O/order_clean.txt
cat $O/order_clean.txt
