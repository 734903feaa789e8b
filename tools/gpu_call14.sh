#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/c14
mkdir -p $O
timeout 400 python bench.py --mode train --batch 8 --steps 5 --warmup 2 > $O/bench_train.log 2>&1; tail -1 $O/bench_train.log > $O/r02_bench_line_train.json
timeout 500 rocprofv3 --kernel-trace --stats -d $O/kt_train -o kt -- python bench.py --mode train --batch 8 --steps 2 --warmup 1 --no-graph > $O/kt_train.log 2>&1
python tools/rocprof_summary.py "$(find $O/kt_train -name '*.db' | head -1)" $O/r02_train_step_kernel_stats.txt > /dev/null
rm -rf $O/kt_train
cat $O/r02_bench_line_train.json; head -16 $O/r02_train_step_kernel_stats.txt
