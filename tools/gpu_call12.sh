#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/c12
mkdir -p $O
export PYTHONFAULTHANDLER=1
timeout 600 python tools/rows_probe.py > $O/rows.log 2>&1
timeout 900 python -m pytest tests/test_configs_gpu.py tests/test_sr_gpu.py -q -m gpu -p no:cacheprovider -s -k "config1 or sr" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
timeout 300 python bench.py --mode sr --batch 1 --steps 1 --warmup 1 > $O/bench_sr.log 2>&1
timeout 200 python bench.py --no-cpu-baseline --no-breakdown > $O/bench.json 2> $O/bench.err
tail -12 $O/rows.log; grep -E "passed|failed|FAILED|Error|configs\[" $O/pytest.log | tail; tail -1 $O/bench_sr.log | cut -c1-330; tail -1 $O/bench.json | cut -c1-330
