#!/bin/bash
# round 6, GPU call 26: sweep of MMD_VCONV_BLOCKS (cap of the fused VideoConv's persistent grid), alternating, three passes
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export O=gpurun_out/c26
mkdir -p $O
B="python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-breakdown"
run() { name=$1; shift; env "$@" timeout 400 $B > $O/b_$name.log 2>&1; tail -1 $O/b_$name.log > $O/line_$name.json; }
for rep in 1 2 3; do
run default_$rep X=1
run vc064_$rep MMD_VCONV_BLOCKS=64
run vc088_$rep MMD_VCONV_BLOCKS=88
run vc104_$rep MMD_VCONV_BLOCKS=104
run vc128_$rep MMD_VCONV_BLOCKS=128
run vc160_$rep MMD_VCONV_BLOCKS=160
done
python - <<'PY' > $O/ab_lines.txt
import json, glob, os
for p in sorted(glob.glob(os.environ["O"] + "/line_*.json")):
    try:
        d = json.load(open(p)); print(f"{os.path.basename(p):34s} ms_per_step {d['ms_per_step']:.3f}  value {d['value']:.1f}")
    except Exception as e:
        print(p, "unreadable", e)
PY
cat $O/ab_lines.txt
