#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/c4
mkdir -p $O
export PYTHONFAULTHANDLER=1
for m in raw node flat; do
  MMD_LANE_FORK=$m timeout 120 python tools/lanes_probe.py 2 > $O/probe_$m.log 2>&1; echo "rc=$?" >> $O/probe_$m.log
done
MMD_LANE_FORK=flat timeout 120 python tools/lanes_probe.py 4 > $O/probe_flat4.log 2>&1; echo "rc=$?" >> $O/probe_flat4.log
timeout 300 python -m pytest tests/test_ops_gpu.py -q -m gpu -p no:cacheprovider -k "attention" > $O/pytest_attn.log 2>&1; echo "rc=$?" >> $O/pytest_attn.log
timeout 120 python tools/attn_bench.py > $O/attn_bench.txt 2>&1
for f in $O/probe_*.log; do echo "== $f"; grep -v "^  File\|Extension modules" $f | tail -n 25; done
tail -5 $O/pytest_attn.log; cat $O/attn_bench.txt
