#!/bin/bash
# round 5, GPU call 4: split-K for the ds8 3x3 convs (tests, then same-call A/B: off / 2 / 4 shares, tile 129 / 132, also on the ds4 frames);
# the audio attention of a cross block behind the video stream's proj_out as well (MMD_CROSS_SERIAL=2).
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export O=gpurun_out/c4
mkdir -p $O
export PYTHONFAULTHANDLER=1
timeout 900 python -m pytest tests/test_round5_gpu.py tests/test_model_gpu.py -x -q -s -p no:cacheprovider > $O/pytest_a.txt 2>&1
tail -4 $O/pytest_a.txt
B="python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-breakdown"
run() { name=$1; shift; env "$@" timeout 300 $B > $O/b_$name.log 2>&1; tail -1 $O/b_$name.log > $O/line_$name.json; }
run splitk2_129_a X=1
run splitk0_a MMD_SPLITK=0
run splitk4_129 MMD_SPLITK=4
run splitk2_132 MMD_SPLITK_TILE=132
run splitk4_132 MMD_SPLITK=4 MMD_SPLITK_TILE=132
run splitk2_ds4 MMD_SPLITK_MAX_PIXELS=256
run cross2 MMD_CROSS_SERIAL=2
run splitk2_129_b X=1
run splitk0_b MMD_SPLITK=0
python - <<'PY' > $O/ab_lines.txt
import json, glob, os
for p in sorted(glob.glob(os.environ["O"] + "/line_*.json")):
    try:
        d = json.load(open(p)); print(f"{os.path.basename(p):34s} ms_per_step {d['ms_per_step']:.3f}  value {d['value']:.1f}")
    except Exception as e:
        print(p, "unreadable", e)
PY
cat $O/ab_lines.txt
