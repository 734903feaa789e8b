#!/bin/bash
# Round 3, GPU call 10: DMA attention variants (occupancy 3, split reduction chains, in-wave S(t+1)-before-softmax(t) pipeline)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c10
mkdir -p $O
V=mm-diffusion_amd/lib/variants
D=/tmp/attn_product_outputs
{ echo "## product"; ATTN_BENCH_SAVE=$D ATTN_BENCH_IMPLS=4 timeout 200 python tools/attn_bench.py
  for n in w3 ch pipe pipech; do echo "## attn_$n"; MMD_LIB=$V/libmmd_attn_$n.so ATTN_BENCH_CMP=$D ATTN_BENCH_IMPLS=4 timeout 200 python tools/attn_bench.py; done; } > $O/attn_variants.txt 2>&1
grep -v amdgpu $O/attn_variants.txt | cut -c1-200
MMD_LIB=$V/libmmd_attn_pipe.so timeout 300 python -m pytest tests/test_round3_gpu.py -q -m gpu -x -p no:cacheprovider -k "attn_dma" > $O/pytest_pipe.txt 2>&1; echo "pipe tests rc=$?"; tail -3 $O/pytest_pipe.txt
