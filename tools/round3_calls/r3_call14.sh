#!/bin/bash
# Round 3, GPU call 14: run-to-run determinism of the plan (tile autotune vs record order), then the fixed tests
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c14
mkdir -p $O
{ echo "## mid default"; timeout 200 python tools/determinism_check.py mid
  echo "## full default"; timeout 300 python tools/determinism_check.py full
  echo "## mid tails"; MMD_GN_TAIL=all timeout 200 python tools/determinism_check.py mid; } 2>&1 | grep -v amdgpu > $O/determinism.txt
cat $O/determinism.txt | cut -c1-250
timeout 900 python -m pytest tests/test_sr_gpu.py tests/test_round3_gpu.py tests/test_ops_gpu.py "tests/test_configs_gpu.py::test_config4_dpm_solver_pp_50_evaluations_then_sr_frame_batch" -m gpu -q -s > $O/tests.txt 2>&1
grep -v amdgpu $O/tests.txt | grep "rel-L2\|configs\|passed\|failed\|Error\|error\|assert" | cut -c1-300 | tail -30
