#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/c34
mkdir -p $O
timeout 300 python tools/sync_balance.py 4 2>&1 | grep -v amdgpu | tee $O/sync_balance.txt | cut -c1-220
