#!/bin/bash
# round 6, GPU call 36: weight pack / gradient unpack on LDS tiles: tests, the training suite, the train bench line and its kernel stats
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c36
mkdir -p $O
timeout 900 python -m pytest tests/test_round6_gpu.py tests/test_train_gpu.py tests/test_trainloop_gpu.py tests/test_configs_gpu.py -m gpu -x -q -k "pack or train or grad" > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
for rep in 1 2; do timeout 400 python bench.py --mode train --batch 8 --steps 5 --warmup 2 > $O/train_$rep.log 2>&1; tail -1 $O/train_$rep.log > $O/line_train_$rep.json; done
BTR="python bench.py --mode train --batch 8 --steps 4 --warmup 2"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/kt_train -o kt -- $BTR > $O/kt_train.log 2>&1
python tools/rocprof_summary.py "$(find $O/kt_train -name '*.db' | head -1)" $O/r06_train_step_kernel_stats.txt $O/kernel_stats_train.json "$BTR" > /dev/null
rm -rf $O/kt_train
python - <<'PY'
import json, glob
for p in sorted(glob.glob("gpurun_out/c36/line_train_*.json")):
    d = json.load(open(p)); print(p, d.get("ms_per_step"), d.get("value"))
PY
grep -E "pack_weights|unpack_grads|adamw" $O/r06_train_step_kernel_stats.txt
