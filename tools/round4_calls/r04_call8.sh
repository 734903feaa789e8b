#!/bin/bash
# round 4, GPU call 8: same-call A/B lines of two launch-geometry switches (results do not depend on either)
mkdir -p gpurun_out/c8
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-breakdown > gpurun_out/c8/$tag.json 2> gpurun_out/c8/$tag.err; python -c "import json; d=json.loads(open('gpurun_out/c8/$tag.json').read().strip().splitlines()[-1]); print('$tag', round(d['ms_per_step'],3))"; }
run base A=1
run strip320 MMD_STRIP_BLOCKS=320
run strip640 MMD_STRIP_BLOCKS=640
run strip896 MMD_STRIP_BLOCKS=896
run halo256 MMD_HALO_MIN_PIXELS=256
run base2 A=1
