#!/bin/bash
# round 5, GPU call 11 (timing diagnostics only): does the video chain ever stand waiting for the audio chain?  The replayed plan without the
# "video waits for audio" markers (MMD_SKIP_WAIT=10), and without the opposite ones (=01), against the plan as it is; cross2 again.
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export O=gpurun_out/c11
mkdir -p $O
B="python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-breakdown"
run() { name=$1; shift; env "$@" timeout 300 $B > $O/b_$name.log 2>&1; tail -1 $O/b_$name.log > $O/line_$name.json; }
run default_1 X=1
run nowait_video MMD_SKIP_WAIT=10
run nowait_audio MMD_SKIP_WAIT=01
run default_2 X=1
run nowait_video_2 MMD_SKIP_WAIT=10
python - <<'PY' > $O/ab_lines.txt
import json, glob, os
for p in sorted(glob.glob(os.environ["O"] + "/line_*.json")):
    try:
        d = json.load(open(p)); print(f"{os.path.basename(p):34s} ms_per_step {d['ms_per_step']:.3f}  value {d['value']:.1f}")
    except Exception as e:
        print(p, "unreadable", e)
PY
cat $O/ab_lines.txt
