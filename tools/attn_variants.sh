#!/bin/bash
# Attention experiment builds against the product build (tools/attn_bench.py, the attention shapes of the Landscape model at batch 4).
# The per-128-query MFMA kernel is VALU-bound at head width 64; each variant removes VALU work from its tile loop:
#   rowsum : -DATTN_ROWSUM_MFMA     softmax denominator on the matrix cores (an all-ones A operand) instead of 32 adds per tile
#   bfkv   : -DATTN_BRANCHFREE_KV   K / V staging loads read a zero page past the window instead of sitting under branches
#   pkfma  : -DATTN_PKFMA           the exp2 arguments by v_pk_fma_f32 (two per instruction)
#   all    : the three together     (static VALU count of attn_mfma_kernel<64>: 865 -> 742, MFMAs 32 -> 40)
# Outputs are compared with the product build's (rel-L2: bfkv / pkfma must give 0; rowsum sums bf16-rounded P, ~1e-3 is expected).
#   gpurun -- 'bash tools/attn_variants.sh > gpurun_out/attn_variants.txt'       (~20 s per variant)
cd "$(dirname "$0")/.."
D=/tmp/attn_product_outputs
echo "## product build"; ATTN_BENCH_SAVE=$D ATTN_BENCH_IMPLS=2 python tools/attn_bench.py
for v in "rowsum:-DATTN_ROWSUM_MFMA" "bfkv:-DATTN_BRANCHFREE_KV" "pkfma:-DATTN_PKFMA" "all:-DATTN_ROWSUM_MFMA -DATTN_BRANCHFREE_KV -DATTN_PKFMA"; do
  name=${v%%:*}; flags=${v#*:}
  bash tools/build_variant.sh attn_$name mmd_attn.hip "$flags" > /dev/null
  echo "## $name ($flags)"
  MMD_LIB=mm-diffusion_amd/lib/variants/libmmd_attn_$name.so ATTN_BENCH_CMP=$D ATTN_BENCH_IMPLS=2 timeout 300 python tools/attn_bench.py
done
