#!/bin/bash
# attention: ablation timings of the staged-window kernel + SQ PMC passes of both MFMA attention kernels
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/c6
mkdir -p $O
export ATTN_BENCH_SHAPES=0,1,2
ATTN_BENCH_IMPLS=2,3 timeout 100 python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids | sed 's/^/BASE  /' > $O/ablate.txt
for a in 1 2 3 4; do
  MMD_LIB=$GRAFT_REPO_ROOT/mm-diffusion_amd/lib/variants/libmmd_ats$a.so ATTN_BENCH_IMPLS=3 timeout 100 python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids | sed "s/^/ABL$a  /" >> $O/ablate.txt
done
cat $O/ablate.txt
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > $O/sq_counters.txt
export ATTN_BENCH_IMPLS=2,3
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace -d $O/pmc1 -o p -f csv -- python tools/attn_bench.py > $O/pmc1.log 2>&1
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_MFMA_MOPS_BF16 --kernel-trace -d $O/pmc2 -o p -f csv -- python tools/attn_bench.py > $O/pmc2.log 2>&1
python tools/pmc_sq_summary.py $O/attn_pmc_sq.txt "python tools/attn_bench.py (ds-2 shapes, both MFMA attention kernels)" $O/pmc1 $O/pmc2
cat $O/attn_pmc_sq.txt | head -80
tail -3 $O/pmc1.log $O/pmc2.log
