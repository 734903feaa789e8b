#!/bin/bash
# round 5, GPU call 14: one workgroup per CU for the small strip launches - also for the ds2 launches (256 row blocks: 256 workgroups of 12 chunks
# instead of 512 of 6)?  MMD_STRIP_BLOCKS_BIG = the split target of launches with >= 256 row blocks.
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export O=gpurun_out/c14
mkdir -p $O
B="python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-breakdown"
run() { name=$1; shift; env "$@" timeout 300 $B > $O/b_$name.log 2>&1; tail -1 $O/b_$name.log > $O/line_$name.json; }
run all256_a X=1
run big448_a MMD_STRIP_BLOCKS_BIG=448
run big768_a MMD_STRIP_BLOCKS_BIG=768
run old448 MMD_STRIP_BLOCKS=448 MMD_TCONV_BLOCKS=512
run all256_b X=1
run big448_b MMD_STRIP_BLOCKS_BIG=448
run big768_b MMD_STRIP_BLOCKS_BIG=768
python - <<'PY' > $O/ab_lines.txt
import json, glob, os
for p in sorted(glob.glob(os.environ["O"] + "/line_*.json")):
    try:
        d = json.load(open(p)); print(f"{os.path.basename(p):34s} ms_per_step {d['ms_per_step']:.3f}  value {d['value']:.1f}")
    except Exception as e:
        print(p, "unreadable", e)
PY
cat $O/ab_lines.txt
