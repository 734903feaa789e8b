#!/bin/bash
# round 5, GPU call 7: is there a launch that can run ahead of its producer on the other stream?  One engine, two inputs fed alternately,
# every output bitwise its first occurrence - for the default plan, the round-4 plan switches, and the (off by default) tail experiment;
# then the tails test of test_round3_gpu.py 12 times with its mode printed.
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export O=gpurun_out/c7
mkdir -p $O
export PYTHONFAULTHANDLER=1
S="python tools/alternating_inputs_stress.py"
{
timeout 200 $S mid 1 600
timeout 200 $S mid 2 400
timeout 300 $S full 1 150
timeout 300 $S full 4 60
MMD_CROSS_SERIAL=0 timeout 200 $S mid 1 600
MMD_CROSS_SERIAL=0 MMD_UP_LOWRES=0 MMD_RESAMPLE_STATS=0 MMD_HEAD_GEMM=0 timeout 200 $S mid 1 600
MMD_GN_TAIL=all timeout 200 $S mid 1 600
MMD_GN_TAIL=all timeout 200 $S mid 2 400
MMD_GN_TAIL=all MMD_CROSS_SERIAL=0 timeout 200 $S mid 1 600
MMD_GN_TAIL=all MMD_CROSS_SERIAL=0 MMD_UP_LOWRES=0 MMD_RESAMPLE_STATS=0 timeout 200 $S mid 1 600
} > $O/alternating.txt 2>&1
cat $O/alternating.txt | grep -v amdgpu.ids
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do
  timeout 200 python -m pytest tests/test_round3_gpu.py -q -p no:cacheprovider -k "tails_match" 2>&1 | grep -E "passed|failed|mode " | head -3
done > $O/tails12.txt 2>&1
cat $O/tails12.txt
