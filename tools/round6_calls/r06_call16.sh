#!/bin/bash
# round 6, GPU call 16: weight gradient (wgrad_tr_bf16_kernel) with an XCD-aware block order - the (co tile, ci tile, tap) blocks of one row split
# on ONE XCD's L2 instead of all eight - against the library of the commit before (lib/variants/libmmd_base2.so).
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export O=gpurun_out/c16
mkdir -p $O
BASE=$PWD/mm-diffusion_amd/lib/variants/libmmd_base2.so
timeout 1200 python -m pytest tests -x -q -m gpu -k "wgrad or train_step or backward" > $O/pytest_wgrad.txt 2>&1; tail -3 $O/pytest_wgrad.txt
echo "== base" > $O/wgrad_bench.txt
MMD_LIB=$BASE timeout 300 python tools/wgrad_bench.py >> $O/wgrad_bench.txt 2>&1
echo "== new" >> $O/wgrad_bench.txt
timeout 300 python tools/wgrad_bench.py >> $O/wgrad_bench.txt 2>&1
grep -v amdgpu.ids $O/wgrad_bench.txt | cut -c1-200
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
