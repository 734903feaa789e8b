#!/bin/bash
# round 6, GPU call 15: the software-pipelined row-strip GEMM (MFMAs of sub-tile s with the epilogue steps of sub-tile s - 1 between them; K >= 256
# instances) + residual rows of the tile kernels' epilogues requested ahead - against the library of the last commit (lib/variants/libmmd_base.so).
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export O=gpurun_out/c15
mkdir -p $O
BASE=$PWD/mm-diffusion_amd/lib/variants/libmmd_base.so
timeout 1200 python -m pytest tests -x -q -m gpu -k "strip or gemm or gn_conv or conv1x1 or halo or resblock or conv" > $O/pytest_strip.txt 2>&1; tail -5 $O/pytest_strip.txt
echo "== base" > $O/strip_probe.txt
MMD_LIB=$BASE timeout 300 python tools/strip_probe.py >> $O/strip_probe.txt 2>&1
echo "== new" >> $O/strip_probe.txt
timeout 300 python tools/strip_probe.py >> $O/strip_probe.txt 2>&1
grep -v amdgpu.ids $O/strip_probe.txt | cut -c1-250
B="python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-breakdown"
run() { name=$1; shift; env "$@" timeout 400 $B > $O/b_$name.log 2>&1; tail -1 $O/b_$name.log > $O/line_$name.json; }
for rep in 1 2 3; do
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
