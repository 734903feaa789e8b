#!/bin/bash
# Round 3, GPU call 44: the model-level parity tests with their per-case bounds
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c44
timeout 600 python -m pytest tests/test_model_gpu.py tests/test_sampling_api_gpu.py -m gpu -q -s 2>&1 | grep -v amdgpu | grep "rel-L2\|passed\|failed\|Error" | tee gpurun_out/c44/measured.txt | grep "ddim\|cond\|guided\|passed\|failed\|Error" | cut -c1-200
