#!/bin/bash
# round 5, GPU call 8: stale-memory probe (two engines in lockstep from different pool fills) of the mid-size model, default plan and the tail
# experiment, batch 1 and 2 - the class of bug a long process would expose in the batch-row test that failed once in 17 runs.
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export O=gpurun_out/c8
mkdir -p $O
{
timeout 300 python tools/stale_read_probe.py mid 1
timeout 300 python tools/stale_read_probe.py mid 2
MMD_GN_TAIL=all timeout 300 python tools/stale_read_probe.py mid 1
MMD_GN_TAIL=all timeout 300 python tools/stale_read_probe.py mid 2
timeout 300 python tools/stale_read_probe.py tiny 2
} > $O/stale.txt 2>&1
grep -v amdgpu.ids $O/stale.txt
