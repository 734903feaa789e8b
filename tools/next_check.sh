#!/bin/bash
# bounded GPU check of the next-round branch (fuse rule + explicit-tile gate): the tests the rule touches
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/next
timeout 95 python -m pytest tests/test_strip_gpu.py tests/test_ops_gpu.py tests/test_model_gpu.py -x -q -p no:cacheprovider -k "strip or gn or norm or model or full_config or rows or graph" > gpurun_out/next/tests.txt 2>&1
echo "rc=$?" >> gpurun_out/next/tests.txt
tail -5 gpurun_out/next/tests.txt
