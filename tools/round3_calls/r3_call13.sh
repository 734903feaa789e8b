#!/bin/bash
# Round 3, GPU call 13: the parity evidence the round-2 review asked for (250-step drift, head width 192 in fp32 mode, configs[3] vs the
# reference fixture) and the two-rank legs of bench.py on one GPU
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c13
mkdir -p $O
timeout 1500 python -m pytest tests/test_sr_gpu.py tests/test_multirank_gpu.py tests/test_configs_gpu.py tests/test_round3_gpu.py -m gpu -q -s > $O/tests.txt 2>&1
grep -v amdgpu $O/tests.txt | grep "rel-L2\|configs\|passed\|failed\|Error\|error\|assert" | cut -c1-300 | tail -50
