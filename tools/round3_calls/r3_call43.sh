#!/bin/bash
# Round 3, GPU call 43: measured distances of the model-level parity tests (to state per-case bounds instead of blanket ones)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c43
timeout 600 python -m pytest tests/test_model_gpu.py tests/test_train_gpu.py tests/test_sr_gpu.py -m gpu -q -s 2>&1 | grep -v amdgpu | grep "rel-L2\|passed\|failed" | tee gpurun_out/c43/measured.txt | cut -c1-200
