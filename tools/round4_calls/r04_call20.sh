#!/bin/bash
# round 4, GPU call 20: which levels use the fused temporal-attention block (same-call A/B)
mkdir -p gpurun_out/c20
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-breakdown > gpurun_out/c20/$tag.json 2> gpurun_out/c20/$tag.err; python -c "import json; d=json.loads(open('gpurun_out/c20/$tag.json').read().strip().splitlines()[-1]); print('$tag', round(d['ms_per_step'],3))" || tail -5 gpurun_out/c20/$tag.err; }
run ds2 MMD_TATTN_LEVELS=256
run ds2ds4 MMD_TATTN_LEVELS=256,384
run ds2b MMD_TATTN_LEVELS=256
run ds2ds4b MMD_TATTN_LEVELS=256,384
