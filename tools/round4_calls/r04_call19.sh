#!/bin/bash
# round 4, GPU call 19: mmd_tattn_block generalised to 384 / 512 channels (head widths 96 / 128: the ds4 / ds8 levels)
mkdir -p gpurun_out/c19
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONFAULTHANDLER=1
timeout 900 python -m pytest tests/test_tattn_gpu.py -x -q -m gpu -p no:cacheprovider > gpurun_out/c19/pytest.txt 2>&1
tail -12 gpurun_out/c19/pytest.txt | cut -c1-250
timeout 300 python tools/tattn_bench.py > gpurun_out/c19/tattn_bench.txt 2>&1; tail -4 gpurun_out/c19/tattn_bench.txt
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-breakdown > gpurun_out/c19/$tag.json 2> gpurun_out/c19/$tag.err; python -c "import json; d=json.loads(open('gpurun_out/c19/$tag.json').read().strip().splitlines()[-1]); print('$tag', round(d['ms_per_step'],3))" || tail -5 gpurun_out/c19/$tag.err; }
run all A=1
run ds2only MMD_TATTN_LEVELS=256
run all2 A=1
run ds2only2 MMD_TATTN_LEVELS=256
