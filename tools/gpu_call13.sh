#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/c13
mkdir -p $O
timeout 400 python tools/train_host_probe.py 8 > $O/probe.log 2>&1
tail -60 $O/probe.log
