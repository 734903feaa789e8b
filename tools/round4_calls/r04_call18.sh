#!/bin/bash
# round 4, GPU call 18: mmd_tconv (temporal conv with stationary activations): tests, micro-benchmark, same-call A/B bench lines
mkdir -p gpurun_out/c18
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONFAULTHANDLER=1
timeout 600 python -m pytest tests/test_tconv_gpu.py -x -q -m gpu -p no:cacheprovider > gpurun_out/c18/pytest.txt 2>&1
tail -25 gpurun_out/c18/pytest.txt | cut -c1-300
timeout 300 python tools/tconv_bench.py > gpurun_out/c18/tconv_bench.txt 2>&1; tail -4 gpurun_out/c18/tconv_bench.txt
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-breakdown > gpurun_out/c18/$tag.json 2> gpurun_out/c18/$tag.err; python -c "import json; d=json.loads(open('gpurun_out/c18/$tag.json').read().strip().splitlines()[-1]); print('$tag', round(d['ms_per_step'],3))" || tail -5 gpurun_out/c18/$tag.err; }
run tconv A=1
run notconv MMD_TCONV=0
run tconv2 A=1
run notconv2 MMD_TCONV=0
