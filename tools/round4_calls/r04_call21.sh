#!/bin/bash
# round 4, GPU call 21: SQ counters of mmd_tconv on tools/tconv_bench.py (no code change: evidence only)
mkdir -p gpurun_out/c21
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python tools/tconv_bench.py"
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace -d gpurun_out/c21/pmc_sq1 -o p -f csv -- $B > gpurun_out/c21/pmc_sq1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA --kernel-trace -d gpurun_out/c21/pmc_sq2 -o p -f csv -- $B > gpurun_out/c21/pmc_sq2.log 2>&1
python tools/pmc_sq_summary.py gpurun_out/c21/tconv_pmc_sq.txt "tools/tconv_bench.py (means over its four shapes: ds2 / ds4 / ds8 at batch 4, ds2 at batch 1)" gpurun_out/c21/pmc_sq1 gpurun_out/c21/pmc_sq2
rm -rf gpurun_out/c21/pmc_sq1 gpurun_out/c21/pmc_sq2
grep -A19 "tconv_kernel" gpurun_out/c21/tconv_pmc_sq.txt | head -70
