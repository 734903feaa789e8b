#!/bin/bash
# One gpurun that regenerates everything under profiles/ for a round AT THE CURRENT BUILD: GPU test suite, smoke, the bench lines of every
# mode, the rocprofv3 kernel trace, the two HBM PMC passes (-> profiles/pmc_traffic.json, stamped with the build id bench.py checks), the
# SQ PMC passes (MFMA busy / wait / stall per kernel) and the final bench line that USES the fresh pmc_traffic.json.
# usage: gpurun -- 'bash tools/round_profile.sh <tag>'      (the LAST GPU call of a round; no csrc/ commit after it)
set -x
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG
mkdir -p $O profiles
export PYTHONFAULTHANDLER=1
if [ -z "$RP_SKIP_TESTS" ]; then      # RP_SKIP_TESTS=1: profiling passes only (the suite was just run at this build)
timeout 2100 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $O/pytest_full.txt 2>&1
tail -3 $O/pytest_full.txt > $O/pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1
fi
B="python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-breakdown"
# the kernel trace ranks the REPLAYED step: 60 steps (40 k launches) beside the ~1 k plan-time autotune launches of the warm-up
BT="python bench.py --steps 60 --warmup 2 --no-cpu-baseline --no-breakdown"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- $BT > $O/kt.log 2>&1
# the same trace as csv, summarised per LAUNCH SHAPE (kernel, grid) and per HW queue (round 5: tools/kt_by_shape.py)
BC="python bench.py --steps 24 --warmup 2 --no-cpu-baseline --no-breakdown"
timeout 600 rocprofv3 --kernel-trace -f csv -d $O/ktc -o kt -- $BC > $O/ktc.log 2>&1
python tools/kt_by_shape.py "$(find $O/ktc -name '*kernel_trace.csv' | head -1)" profiles/${TAG}_kernel_trace_by_shape.txt 20 > /dev/null
timeout 500 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch -o p -f csv -- $B > $O/pmc_fetch.log 2>&1
timeout 500 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write -o p -f csv -- $B > $O/pmc_write.log 2>&1
timeout 500 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace -d $O/pmc_sq1 -o p -f csv -- $B > $O/pmc_sq1.log 2>&1
timeout 500 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA --kernel-trace -d $O/pmc_sq2 -o p -f csv -- $B > $O/pmc_sq2.log 2>&1
python tools/pmc_summary.py $O/pmc_fetch $O/pmc_write profiles/${TAG}_pmc_hbm.txt profiles/pmc_traffic.json "$B"
python tools/pmc_sq_summary.py profiles/${TAG}_pmc_sq.txt "$B" $O/pmc_sq1 $O/pmc_sq2
python tools/rocprof_summary.py "$(find $O/kt -name '*.db' | head -1)" profiles/${TAG}_kernel_stats_default_bench.txt profiles/kernel_stats.json "$BT" > /dev/null
# the training step (cfg4, batch 8): which kernel families the step's time is in (bench.py --mode train quotes it)
BTR="python bench.py --mode train --batch 8 --steps 4 --warmup 2"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/kt_train -o kt -- $BTR > $O/kt_train.log 2>&1
python tools/rocprof_summary.py "$(find $O/kt_train -name '*.db' | head -1)" profiles/${TAG}_train_step_kernel_stats.txt profiles/kernel_stats_train.json "$BTR" > /dev/null
timeout 600 python bench.py --breakdown-out profiles/${TAG}_bench_breakdown_hip_events.json > $O/bench_default.log 2>&1
tail -1 $O/bench_default.log > profiles/${TAG}_bench_line.json
timeout 400 python bench.py --mode dpm --batch 2 --steps 2 --warmup 1 > $O/bench_dpm.log 2>&1; tail -1 $O/bench_dpm.log > profiles/${TAG}_bench_line_dpm.json
timeout 400 python bench.py --mode sr --batch 1 --steps 1 --warmup 1 > $O/bench_sr.log 2>&1; tail -1 $O/bench_sr.log > profiles/${TAG}_bench_line_sr.json
timeout 400 python bench.py --mode train --batch 8 --steps 5 --warmup 2 > $O/bench_train.log 2>&1; tail -1 $O/bench_train.log > profiles/${TAG}_bench_line_train.json
mkdir -p $O/profiles && cp profiles/${TAG}_* profiles/pmc_traffic.json profiles/kernel_stats.json profiles/kernel_stats_train.json $O/profiles/ 2>/dev/null
rm -rf $O/kt $O/ktc $O/kt_train $O/pmc_fetch $O/pmc_write $O/pmc_sq1 $O/pmc_sq2      # raw traces stay on the box: only the summaries travel back
tail -3 $O/pytest.txt 2>/dev/null; tail -1 $O/smoke.log 2>/dev/null; cat profiles/${TAG}_bench_line.json
