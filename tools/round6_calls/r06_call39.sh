#!/bin/bash
# round 6, GPU call 39: what each launch chain costs the other at the final build (one chain dropped from the replayed plan: timing only),
# with two batch lanes (default) and with one
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export O=gpurun_out/c39
mkdir -p $O
export MMD_TIMING_ONLY=1
run() { name=$1; lanes=$2; shift; shift; env "$@" timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-breakdown --lanes $lanes > $O/b_$name.log 2>&1; tail -1 $O/b_$name.log > $O/line_$name.json; }
run lanes2_both 2 X=1
run lanes2_video_only 2 MMD_SKIP_SID=1
run lanes2_audio_only 2 MMD_SKIP_SID=0
run lanes1_both 1 X=1
run lanes1_video_only 1 MMD_SKIP_SID=1
run lanes1_audio_only 1 MMD_SKIP_SID=0
run lanes2_both_again 2 X=1
python - <<'PY' > $O/ab_lines.txt
import json, glob, os
for p in sorted(glob.glob(os.environ["O"] + "/line_*.json"), key=os.path.getmtime):
    try:
        d = json.load(open(p)); print(f"{os.path.basename(p):34s} ms_per_step {d['ms_per_step']:.3f}")
    except Exception as e:
        print(os.path.basename(p), "unreadable", str(e)[:60])
PY
cat $O/ab_lines.txt
