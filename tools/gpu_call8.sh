#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/c8
mkdir -p $O
export PYTHONFAULTHANDLER=1
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider -s > $O/pytest_full.log 2>&1; echo "rc=$?" >> $O/pytest_full.log
timeout 300 python bench.py --mode sr --batch 1 --steps 1 --warmup 1 > $O/bench_sr.log 2>&1
MMD_SR_GRAPH=0 timeout 300 python bench.py --mode sr --batch 1 --steps 1 --warmup 1 > $O/bench_sr_eager.log 2>&1
timeout 200 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
grep -E "passed|failed|rel-L2|FAILED|Error" $O/pytest_full.log | tail -60
tail -1 $O/bench_sr.log; tail -1 $O/bench_sr_eager.log; tail -1 $O/bench.json
