#!/bin/bash
# round 4, GPU call 16: SQ counters of mmd_tattn_block (workgroup shapes 0 and 1) on tools/tattn_bench.py
mkdir -p gpurun_out/c16
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python tools/tattn_bench.py"
for cfg in 0 1; do
export MMD_TATTN_CFG=$cfg
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace -d gpurun_out/c16/pmc_sq1_$cfg -o p -f csv -- $B > gpurun_out/c16/pmc_sq1_$cfg.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA --kernel-trace -d gpurun_out/c16/pmc_sq2_$cfg -o p -f csv -- $B > gpurun_out/c16/pmc_sq2_$cfg.log 2>&1
python tools/pmc_sq_summary.py gpurun_out/c16/tattn_pmc_sq_cfg$cfg.txt "tools/tattn_bench.py, MMD_TATTN_CFG=$cfg" gpurun_out/c16/pmc_sq1_$cfg gpurun_out/c16/pmc_sq2_$cfg
rm -rf gpurun_out/c16/pmc_sq1_$cfg gpurun_out/c16/pmc_sq2_$cfg
grep -A20 "tattn_kernel" gpurun_out/c16/tattn_pmc_sq_cfg$cfg.txt | head -24
done
