#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/c10
mkdir -p $O
timeout 600 python tools/stale_read_probe.py full 1 > $O/probe_full1.log 2>&1
timeout 300 python tools/stale_read_probe.py mid 2 > $O/probe_mid2.log 2>&1
tail -12 $O/probe_full1.log; tail -8 $O/probe_mid2.log
