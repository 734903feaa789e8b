#!/bin/bash
# Round 3, GPU call 36: DMA + transposing-read weight-gradient kernel - parity, micro-benchmark, training step A/B
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c36
mkdir -p $O
timeout 600 python -m pytest tests/test_bwd_gpu.py tests/test_train_gpu.py -m gpu -x -q > $O/tests.txt 2>&1; tail -5 $O/tests.txt | cut -c1-300
{ echo "## MMD_WGRAD_TR=1"; timeout 200 python tools/wgrad_bench.py; echo "## MMD_WGRAD_TR=0"; MMD_WGRAD_TR=0 timeout 200 python tools/wgrad_bench.py; } 2>&1 | grep -v amdgpu | tee $O/wgrad_bench.txt
for v in 1 0 1 0; do MMD_WGRAD_TR=$v timeout 300 python bench.py --mode train --batch 8 --steps 5 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('MMD_WGRAD_TR=$v', round(r['ms_per_step'],2), 'ms/step')"; done | tee $O/train_ab.txt
