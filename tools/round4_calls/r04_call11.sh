#!/bin/bash
# round 4, GPU call 11: consumer-side GroupNorm finalize split by consumer kind (gn_apply_rec only / strip only), same-call A/B
mkdir -p gpurun_out/c11
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-breakdown > gpurun_out/c11/$tag.json 2> gpurun_out/c11/$tag.err; python -c "import json; d=json.loads(open('gpurun_out/c11/$tag.json').read().strip().splitlines()[-1]); print('$tag', round(d['ms_per_step'],3))"; }
run norec MMD_GN_REC=0
run apply_only MMD_GN_REC_STRIP=0 MMD_GN_REC_MAX_ROWS=1000000
run apply_only16k MMD_GN_REC_STRIP=0
run strip_only16k MMD_GN_REC_APPLY=0
run norec2 MMD_GN_REC=0
run apply_only2 MMD_GN_REC_STRIP=0 MMD_GN_REC_MAX_ROWS=1000000
run apply_only16k2 MMD_GN_REC_STRIP=0
run strip_only16k2 MMD_GN_REC_APPLY=0
