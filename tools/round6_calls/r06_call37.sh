#!/bin/bash
# round 6, GPU call 37: GroupNorm backward on a kept-zero workspace (mmd_gn_bwd_ws0): tests, training suites, train step A/B (alternating)
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c37
mkdir -p $O
timeout 900 python -m pytest tests/test_round6_gpu.py tests/test_train_gpu.py tests/test_trainloop_gpu.py tests/test_configs_gpu.py tests/test_sampling_api_gpu.py -m gpu -x -q -k "gn_bwd or pack or train or grad or guided" > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
for rep in 1 2 3; do
for v in 0 1; do MMD_GN_BWD_WS0=$v timeout 400 python bench.py --mode train --batch 8 --steps 5 --warmup 2 > $O/train_ws${v}_$rep.log 2>&1; tail -1 $O/train_ws${v}_$rep.log > $O/line_ws${v}_$rep.json; done
done
python - <<'PY'
import json, glob
for p in sorted(glob.glob("gpurun_out/c37/line_ws*.json")):
    try:
        d = json.load(open(p)); print(p, round(d.get("ms_per_step"), 3))
    except Exception as e:
        print(p, "unreadable", e)
PY
