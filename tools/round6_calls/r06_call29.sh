#!/bin/bash
# round 6, GPU call 29: mmd_gn_group (one-launch GroupNorm for the 400-row audio slices at ds8): tests, model-level parity, step A/B
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export O=gpurun_out/c29
mkdir -p $O
timeout 900 python -m pytest tests/test_round6_gpu.py tests/test_model_gpu.py tests/test_lifetime_gpu.py tests/test_round3_gpu.py tests/test_ops_gpu.py -m gpu -x -q > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
B="python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-breakdown"
run() { name=$1; shift; env "$@" timeout 400 $B > $O/b_$name.log 2>&1; tail -1 $O/b_$name.log > $O/line_$name.json; }
for rep in 1 2 3; do
run group0_$rep MMD_GN_GROUP=0
run group1_$rep MMD_GN_GROUP=1
done
python - <<'PY' > $O/ab_lines.txt
import json, glob, os
for p in sorted(glob.glob(os.environ["O"] + "/line_*.json")):
    try:
        d = json.load(open(p))
        print(f"{os.path.basename(p):34s} ms_per_step {d['ms_per_step']:.3f}  value {d['value']:.1f}")
    except Exception as e:
        print(p, "unreadable", e)
PY
cat $O/ab_lines.txt
