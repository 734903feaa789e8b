#!/bin/bash
# round 5, GPU call 12: WHERE inside a video ResBlock the audio chain's ResBlock starts (MMD_AUDIO_HOLD: behind the video chain's prologue =
# in front of its 3x3 conv / behind the conv / behind the temporal conv; at the small levels only or everywhere), and how many workgroups the
# row-strip GEMM spreads a small launch over (MMD_STRIP_BLOCKS) - call 11 showed that the step is 8.5 ms when the audio chain does not have to
# run beside the video chain's latency-bound launches.
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export O=gpurun_out/c12
mkdir -p $O
B="python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-breakdown"
run() { name=$1; shift; env "$@" timeout 300 $B > $O/b_$name.log 2>&1; tail -1 $O/b_$name.log > $O/line_$name.json; }
run default_1 X=1
run hold_pre_all MMD_AUDIO_HOLD=pre
run hold_pre_small MMD_AUDIO_HOLD=pre MMD_AUDIO_HOLD_MAXPIX=256
run hold_pre_ds2 MMD_AUDIO_HOLD=pre MMD_AUDIO_HOLD_MAXPIX=1024
run hold_post_all MMD_AUDIO_HOLD=post
run hold_post_small MMD_AUDIO_HOLD=post MMD_AUDIO_HOLD_MAXPIX=256
run hold_post_ds2 MMD_AUDIO_HOLD=post MMD_AUDIO_HOLD_MAXPIX=1024
run hold_tconv_all MMD_AUDIO_HOLD=tconv
run hold_tconv_small MMD_AUDIO_HOLD=tconv MMD_AUDIO_HOLD_MAXPIX=256
run default_2 X=1
run strip128 MMD_STRIP_BLOCKS=128
run strip192 MMD_STRIP_BLOCKS=192
run strip256 MMD_STRIP_BLOCKS=256
run strip320 MMD_STRIP_BLOCKS=320
run default_3 X=1
python - <<'PY' > $O/ab_lines.txt
import json, glob, os
for p in sorted(glob.glob(os.environ["O"] + "/line_*.json")):
    try:
        d = json.load(open(p)); print(f"{os.path.basename(p):34s} ms_per_step {d['ms_per_step']:.3f}  value {d['value']:.1f}")
    except Exception as e:
        print(p, "unreadable", e)
PY
cat $O/ab_lines.txt
