#!/bin/bash
# round 4, GPU call 15: mmd_tattn_block workgroup shapes: 128 rows x 2 per CU (default) / 256 rows x 8 waves / 256 rows x 4 waves
mkdir -p gpurun_out/c15
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONFAULTHANDLER=1
timeout 600 python -m pytest tests/test_tattn_gpu.py -x -q -m gpu -p no:cacheprovider > gpurun_out/c15/pytest.txt 2>&1
tail -4 gpurun_out/c15/pytest.txt
for cfg in 0 1 2; do MMD_TATTN_CFG=$cfg timeout 300 python tools/tattn_bench.py > gpurun_out/c15/tattn_bench_cfg$cfg.txt 2>&1; echo cfg $cfg; tail -3 gpurun_out/c15/tattn_bench_cfg$cfg.txt; done
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-breakdown > gpurun_out/c15/$tag.json 2> gpurun_out/c15/$tag.err; python -c "import json; d=json.loads(open('gpurun_out/c15/$tag.json').read().strip().splitlines()[-1]); print('$tag', round(d['ms_per_step'],3))" || tail -5 gpurun_out/c15/$tag.err; }
run cfg0 A=1
run cfg1 MMD_TATTN_CFG=1
run unfused MMD_TATTN_FUSED=0
run cfg0b A=1
