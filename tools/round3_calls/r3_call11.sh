#!/bin/bash
# Round 3, GPU call 11: lean-register attention (4 waves per SIMD), tile comparison per layer shape after the issue-side diet
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c11
mkdir -p $O
V=mm-diffusion_amd/lib/variants
D=/tmp/attn_product_outputs
{ echo "## product"; ATTN_BENCH_SAVE=$D ATTN_BENCH_IMPLS=4 timeout 200 python tools/attn_bench.py
  for n in lean4 lean3; do echo "## attn_$n"; MMD_LIB=$V/libmmd_attn_$n.so ATTN_BENCH_CMP=$D ATTN_BENCH_IMPLS=4 timeout 200 python tools/attn_bench.py; done; } > $O/attn_variants.txt 2>&1
grep -v amdgpu $O/attn_variants.txt | cut -c1-200
timeout 300 python tools/gemm_bench.py > $O/gemm_bench.txt 2>&1; grep -v amdgpu $O/gemm_bench.txt | cut -c1-330
