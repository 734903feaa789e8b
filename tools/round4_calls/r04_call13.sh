#!/bin/bash
# round 4, GPU call 13: mmd_tattn_block with 8 waves x 32 rows (RF = 1, default) against 4 waves x 64 rows (MMD_TATTN_RF=2)
mkdir -p gpurun_out/c13
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONFAULTHANDLER=1
timeout 600 python -m pytest tests/test_tattn_gpu.py -x -q -m gpu -p no:cacheprovider > gpurun_out/c13/pytest.txt 2>&1
tail -8 gpurun_out/c13/pytest.txt
MMD_TATTN_RF=2 timeout 600 python -m pytest tests/test_tattn_gpu.py -x -q -m gpu -p no:cacheprovider > gpurun_out/c13/pytest_rf2.txt 2>&1
tail -3 gpurun_out/c13/pytest_rf2.txt
timeout 300 python tools/tattn_bench.py > gpurun_out/c13/tattn_bench.txt 2>&1; tail -3 gpurun_out/c13/tattn_bench.txt
MMD_TATTN_RF=2 timeout 300 python tools/tattn_bench.py > gpurun_out/c13/tattn_bench_rf2.txt 2>&1; tail -3 gpurun_out/c13/tattn_bench_rf2.txt
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-breakdown > gpurun_out/c13/$tag.json 2> gpurun_out/c13/$tag.err; python -c "import json; d=json.loads(open('gpurun_out/c13/$tag.json').read().strip().splitlines()[-1]); print('$tag', round(d['ms_per_step'],3))" || tail -5 gpurun_out/c13/$tag.err; }
run rf1 A=1
run rf2 MMD_TATTN_RF=2
run unfused MMD_TATTN_FUSED=0
run rf1b A=1
