#!/bin/bash
# round 4, GPU call 6: attention with v_permlane32_swap reductions; SQ counters of the attention kernels on the RS cross-attention shape
mkdir -p gpurun_out/c6
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_round3_gpu.py -x -q -k "attn or attention" > gpurun_out/c6/pytest_attn.txt 2>&1; echo "rc=$?" >> gpurun_out/c6/pytest_attn.txt
tail -4 gpurun_out/c6/pytest_attn.txt
timeout 300 python tools/attn_bench.py > gpurun_out/c6/attn_bench.txt 2>&1
cat gpurun_out/c6/attn_bench.txt
B="python tools/attn_bench.py"
export ATTN_BENCH_SHAPES=0,1,2 ATTN_BENCH_IMPLS=4,5,6
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace -d gpurun_out/c6/pmc_sq1 -o p -f csv -- $B > gpurun_out/c6/pmc_sq1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA --kernel-trace -d gpurun_out/c6/pmc_sq2 -o p -f csv -- $B > gpurun_out/c6/pmc_sq2.log 2>&1
python tools/pmc_sq_summary.py gpurun_out/c6/attn_pmc_sq.txt "tools/attn_bench.py shapes 0-2 (spatial ds2, v<-a ds2, a<-v ds2), impl 4 / 5 / 6" gpurun_out/c6/pmc_sq1 gpurun_out/c6/pmc_sq2
rm -rf gpurun_out/c6/pmc_sq1 gpurun_out/c6/pmc_sq2
cat gpurun_out/c6/attn_pmc_sq.txt | head -80
