#!/bin/bash
# round 4, GPU call 12: the fused temporal-attention block (mmd_tattn_block): its tests, the micro-benchmark, same-call A/B bench lines
mkdir -p gpurun_out/c12
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONFAULTHANDLER=1
timeout 600 python -m pytest tests/test_tattn_gpu.py -x -q -m gpu -p no:cacheprovider > gpurun_out/c12/pytest.txt 2>&1
tail -25 gpurun_out/c12/pytest.txt
timeout 300 python tools/tattn_bench.py > gpurun_out/c12/tattn_bench.txt 2>&1; tail -5 gpurun_out/c12/tattn_bench.txt
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-breakdown > gpurun_out/c12/$tag.json 2> gpurun_out/c12/$tag.err; python -c "import json; d=json.loads(open('gpurun_out/c12/$tag.json').read().strip().splitlines()[-1]); print('$tag', round(d['ms_per_step'],3))" || tail -5 gpurun_out/c12/$tag.err; }
run fused A=1
run unfused MMD_TATTN_FUSED=0
run fused2 A=1
run unfused2 MMD_TATTN_FUSED=0
