#!/bin/bash
# does the default bench (20 steps after 3 warm-up steps) sit on a clock ramp?  same box, back to back
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c37
mkdir -p $O
for a in "--steps 20 --warmup 3" "--steps 30 --warmup 3" "--steps 20 --warmup 20" "--steps 60 --warmup 20" "--steps 20 --warmup 3" "--steps 60 --warmup 20" "--steps 250 --warmup 20"; do
  timeout 300 python bench.py $a --no-cpu-baseline --no-breakdown 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$a', round(r['ms_per_step'],3))"
done | tee $O/warmup.txt
