#!/bin/bash
# One gpurun that regenerates everything under profiles/ for a round: GPU test suite, smoke, the bench lines of every mode,
# rocprofv3 kernel trace and the two PMC passes of the default bench.  usage: gpurun -- 'bash tools/round_profile.sh <tag>'
set -x
TAG=${1:-f}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG
mkdir -p $O
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 > $O/pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1
timeout 600 python bench.py --breakdown-out $O/breakdown.json > $O/bench_default.log 2>&1
timeout 400 python bench.py --mode dpm --batch 2 --steps 2 --warmup 1 > $O/bench_dpm.log 2>&1
timeout 400 python bench.py --mode sr --batch 1 --steps 1 --warmup 1 > $O/bench_sr.log 2>&1
timeout 400 python bench.py --mode train --batch 8 --steps 5 --warmup 2 > $O/bench_train.log 2>&1
timeout 500 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-breakdown > $O/kt.log 2>&1
timeout 500 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch -o p -f csv -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-breakdown > $O/pmc_fetch.log 2>&1
timeout 500 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write -o p -f csv -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-breakdown > $O/pmc_write.log 2>&1
timeout 500 rocprofv3 --kernel-trace --stats -d $O/kt_train -o kt -- python bench.py --mode train --batch 8 --steps 2 --warmup 1 --no-graph > $O/kt_train.log 2>&1
tail -3 $O/pytest.txt; tail -1 $O/smoke.log
