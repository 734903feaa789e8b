#!/bin/bash
# round 4, GPU call 17: mmd_tattn_block with the spatial block's proj_out + residual as its front stage (MMD_TATTN_PRE)
mkdir -p gpurun_out/c17
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONFAULTHANDLER=1
timeout 600 python -m pytest tests/test_tattn_gpu.py -x -q -m gpu -p no:cacheprovider > gpurun_out/c17/pytest.txt 2>&1
tail -12 gpurun_out/c17/pytest.txt
timeout 300 python tools/tattn_bench.py > gpurun_out/c17/tattn_bench.txt 2>&1; tail -3 gpurun_out/c17/tattn_bench.txt
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-breakdown > gpurun_out/c17/$tag.json 2> gpurun_out/c17/$tag.err; python -c "import json; d=json.loads(open('gpurun_out/c17/$tag.json').read().strip().splitlines()[-1]); print('$tag', round(d['ms_per_step'],3))" || tail -5 gpurun_out/c17/$tag.err; }
run pre A=1
run nopre MMD_TATTN_PRE=0
run pre2 A=1
run nopre2 MMD_TATTN_PRE=0
