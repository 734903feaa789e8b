#!/bin/bash
# round 4, GPU call 10: consumer-side GroupNorm finalize restricted to launches of one round of blocks (row cap), same-call A/B
mkdir -p gpurun_out/c10
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-breakdown > gpurun_out/c10/$tag.json 2> gpurun_out/c10/$tag.err; python -c "import json; d=json.loads(open('gpurun_out/c10/$tag.json').read().strip().splitlines()[-1]); print('$tag', round(d['ms_per_step'],3))"; }
run norec MMD_GN_REC=0
run rows16k A=1
run rows4k MMD_GN_REC_MAX_ROWS=4096
run rows16k_32k MMD_GN_REC_MAX_BYTES=32768
run norec2 MMD_GN_REC=0
run rows16k2 A=1
run rows4k2 MMD_GN_REC_MAX_ROWS=4096
