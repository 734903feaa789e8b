#!/bin/bash
# round 6, GPU call 2: what binds attn_pipe_kernel?  Timing-only ablations of the generated stream (no MFMAs / no VALU / no LDS reads /
# v_mov instead of v_exp / MFMAs only), schedule parameters (MFMA gap 6 / 12, LDS reads 30 instructions ahead), one block per CU,
# and the SQ counters of impl 4 vs impl 5 on the three ds2 shapes.
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export O=gpurun_out/c2
mkdir -p $O
export ATTN_BENCH_SHAPES=0,1,2 ATTN_BENCH_IMPLS=4,5
for rep in 1 2; do
  echo "== product (rep $rep)" >> $O/ablate.txt
  timeout 300 python tools/attn_bench.py >> $O/ablate.txt 2>&1
done
for v in nomfma novalu nolds noexp mfmaonly gap6 gap12 lds30; do
  echo "== variant $v" >> $O/ablate.txt
  MMD_LIB=$PWD/mm-diffusion_amd/lib/variants/libmmd_$v.so ATTN_BENCH_IMPLS=5 timeout 300 python tools/attn_bench.py >> $O/ablate.txt 2>&1
done
echo "== product, one block per CU (MMD_ATTN_PIPE_LDSPAD=65536)" >> $O/ablate.txt
MMD_ATTN_PIPE_LDSPAD=65536 ATTN_BENCH_IMPLS=5 timeout 300 python tools/attn_bench.py >> $O/ablate.txt 2>&1
echo "== product, three blocks per CU impossible (192 VGPRs); LDS pad 20480 = still two" >> $O/ablate.txt
grep -v amdgpu.ids $O/ablate.txt | sed 's/| dma-exact.*//' > $O/ablate_clean.txt
cat $O/ablate_clean.txt
B="python tools/attn_bench.py"
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace -d $O/pmc_sq1 -o p -f csv -- $B > $O/pmc_sq1.log 2>&1
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA --kernel-trace -d $O/pmc_sq2 -o p -f csv -- $B > $O/pmc_sq2.log 2>&1
python tools/pmc_sq_summary.py $O/attn_pmc_sq.txt "tools/attn_bench.py shapes 0-2 (spatial ds2, v<-a ds2, a<-v ds2), impl 4 / 5" $O/pmc_sq1 $O/pmc_sq2
grep -A20 "attn_pipe_kernel\|attn_dma_kernel" $O/attn_pmc_sq.txt | head -60
rm -rf $O/pmc_sq1 $O/pmc_sq2
