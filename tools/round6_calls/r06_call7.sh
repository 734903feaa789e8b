#!/bin/bash
# round 6, GPU call 7 (timing experiment): the audio chain on its own compute units.  Eager launches (a captured graph does not keep a
# stream's CU mask): the audio chain's stream created with hipExtStreamCreateWithCUMask over 32 .. 96 CUs in three bit patterns (the CU
# numbering of the mask is not documented), the video chain on an unmasked stream or on the complement.
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export O=gpurun_out/c7
mkdir -p $O
B="python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-breakdown"
run() { name=$1; shift; env "$@" timeout 300 $B "${EXTRA[@]}" > $O/b_$name.log 2>&1; tail -1 $O/b_$name.log > $O/line_$name.json; }
EXTRA=()
run graph_1 X=1
EXTRA=(--no-graph)
run eager_1 X=1
run eager_low64 MMD_AUX_CU_MASK=low:64
run eager_spread64 MMD_AUX_CU_MASK=spread:64
run eager_xcd64 MMD_AUX_CU_MASK=xcd:64
run eager_xcd32 MMD_AUX_CU_MASK=xcd:32
run eager_xcd96 MMD_AUX_CU_MASK=xcd:96
run eager_spread32 MMD_AUX_CU_MASK=spread:32
run eager_spread128 MMD_AUX_CU_MASK=spread:128
run eager_low64_rest MMD_AUX_CU_MASK=low:64 MMD_SIDE_CU_MASK=notlow:64
run eager_2 X=1
EXTRA=()
run graph_2 X=1
python - <<'PY' > $O/ab_lines.txt
import json, glob, os
for p in sorted(glob.glob(os.environ["O"] + "/line_*.json")):
    try:
        d = json.load(open(p)); print(f"{os.path.basename(p):34s} ms_per_step {d['ms_per_step']:.3f}  value {d['value']:.1f}")
    except Exception as e:
        print(p, "unreadable", e, open(p.replace('line_', 'b_').replace('.json', '.log')).read()[-400:])
PY
cat $O/ab_lines.txt
