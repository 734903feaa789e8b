#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/c9
mkdir -p $O
export PYTHONFAULTHANDLER=1
timeout 1200 python -m pytest tests/test_configs_gpu.py tests/test_sr_gpu.py tests/test_model_gpu.py -q -m gpu -p no:cacheprovider -s > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_sr -o kt -- python bench.py --mode sr --batch 1 --steps 1 --warmup 1 > $O/kt_sr.log 2>&1
python tools/rocprof_summary.py "$(find $O/kt_sr -name '*.db' | head -1)" $O/sr_kernel_stats.txt > /dev/null 2>&1
timeout 200 python bench.py --no-cpu-baseline --no-breakdown > $O/bench.json 2> $O/bench.err
grep -E "passed|failed|configs\[|FAILED|Error|rows" $O/pytest.log | tail -30
head -30 $O/sr_kernel_stats.txt
tail -1 $O/kt_sr.log | cut -c1-300; tail -1 $O/bench.json | cut -c1-400
