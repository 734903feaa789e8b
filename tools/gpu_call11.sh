#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/c11
mkdir -p $O
timeout 600 python tools/rows_probe.py > $O/rows.log 2>&1
MMD_GN_EPILOGUE=0 timeout 600 python tools/rows_probe.py > $O/rows_noepi.log 2>&1
tail -16 $O/rows.log; echo ---; tail -16 $O/rows_noepi.log
