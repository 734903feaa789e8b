#!/bin/bash
# round 6, GPU call 22: GroupNorm backward in three launches instead of four (the per-slice parameter stage folded into the apply kernel) - backward
# tests and training step A/B against the library of the commit before (lib/variants/libmmd_base5.so).
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export O=gpurun_out/c22
mkdir -p $O
BASE=$PWD/mm-diffusion_amd/lib/variants/libmmd_base5.so
timeout 1500 python -m pytest tests/test_bwd_gpu.py tests/test_train_gpu.py tests/test_configs_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
B="python bench.py --mode train --batch 8 --no-cpu-baseline --no-breakdown"
run() { name=$1; shift; env "$@" timeout 600 $B > $O/b_$name.log 2>&1; tail -1 $O/b_$name.log > $O/line_$name.json; }
for rep in 1 2; do
run base_$rep MMD_LIB=$BASE
run new_$rep X=1
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
