#!/bin/bash
# Attention experiment builds against the product build (tools/attn_bench.py, the attention shapes of the Landscape model at batch 4):
#   rowsum : -DATTN_ROWSUM_MFMA  softmax denominator on the matrix cores (an all-ones A operand) instead of 32 VALU adds per tile
# The variant's outputs are compared with the product build's (rel-L2; the sum runs over bf16-rounded P, so ~1e-3 is expected, not 0).
#   gpurun -- 'bash tools/attn_variants.sh > gpurun_out/attn_variants.txt'
cd "$(dirname "$0")/.."
D=/tmp/attn_product_outputs
echo "## product build"; ATTN_BENCH_SAVE=$D python tools/attn_bench.py
bash tools/build_variant.sh attn_rowsum mmd_attn.hip "-DATTN_ROWSUM_MFMA" > /dev/null
echo "## ATTN_ROWSUM_MFMA"
MMD_LIB=mm-diffusion_amd/lib/variants/libmmd_attn_rowsum.so ATTN_BENCH_CMP=$D timeout 300 python tools/attn_bench.py
