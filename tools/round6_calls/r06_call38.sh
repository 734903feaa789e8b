#!/bin/bash
# round 6, GPU call 38: parallel gn_bwd parameter stage (grid 4 x S) + mmd_gn_group in the training forward (MMD_GN_GROUP A/B): tests, train step alternating
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c38
mkdir -p $O
timeout 900 python -m pytest tests/test_round6_gpu.py tests/test_train_gpu.py tests/test_trainloop_gpu.py tests/test_configs_gpu.py tests/test_sampling_api_gpu.py -m gpu -x -q -k "gn_bwd or gn_group or pack or train or grad or guided" > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
for rep in 1 2 3; do
for v in 0 1; do MMD_GN_GROUP=$v timeout 400 python bench.py --mode train --batch 8 --steps 5 --warmup 2 > $O/train_gg${v}_$rep.log 2>&1; tail -1 $O/train_gg${v}_$rep.log > $O/line_gg${v}_$rep.json; done
done
python - <<'PY'
import json, glob
for p in sorted(glob.glob("gpurun_out/c38/line_gg*.json")):
    try:
        d = json.load(open(p)); print(p, round(d.get("ms_per_step"), 3))
    except Exception as e:
        print(p, "unreadable", e)
PY
