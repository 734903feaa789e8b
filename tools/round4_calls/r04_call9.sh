#!/bin/bash
# round 4, GPU call 9: GroupNorm finalised in the consumer (mmd_gn_apply_rec / mmd_gn_conv1x1_rec) - its tests, the neighbouring suites,
# and same-call A/B bench lines (MMD_GN_REC=0: the finalize launches) with the record bound varied
mkdir -p gpurun_out/c9
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gnrec_gpu.py tests/test_strip_gpu.py tests/test_round3_gpu.py tests/test_model_gpu.py -x -q -m gpu -p no:cacheprovider > gpurun_out/c9/pytest.txt 2>&1
tail -15 gpurun_out/c9/pytest.txt
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-breakdown > gpurun_out/c9/$tag.json 2> gpurun_out/c9/$tag.err; python -c "import json; d=json.loads(open('gpurun_out/c9/$tag.json').read().strip().splitlines()[-1]); print('$tag', round(d['ms_per_step'],3))"; }
run rec A=1
run norec MMD_GN_REC=0
run rec32k MMD_GN_REC_MAX_BYTES=32768
run rec128k MMD_GN_REC_MAX_BYTES=131072
run rec2 A=1
run norec2 MMD_GN_REC=0
