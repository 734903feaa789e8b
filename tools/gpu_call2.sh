#!/bin/bash
# round 2, call 2: root-cause probe on the round-1 tree, then the fixed tree: probe, full GPU suite, halo experiment (incl. base bench)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/c2
mkdir -p $O
export PYTHONFAULTHANDLER=1
(cd _head_copy && timeout 300 python -X faulthandler $GRAFT_REPO_ROOT/tools/diag_gc_capture.py 3 > $O/probe_round1_tree.log 2>&1; echo "rc=$?" >> $O/probe_round1_tree.log)
timeout 300 python -X faulthandler tools/diag_gc_capture.py 3 > $O/probe_fixed_tree.log 2>&1; echo "rc=$?" >> $O/probe_fixed_tree.log
timeout 1200 python -X faulthandler -m pytest tests -q -m gpu -p no:cacheprovider > $O/pytest_full.log 2>&1; echo "rc=$?" >> $O/pytest_full.log
bash tools/try_halo.sh > $O/try_halo.log 2>&1
for f in $O/probe_round1_tree.log $O/probe_fixed_tree.log $O/pytest_full.log; do echo "== $f"; tail -n 8 $f; done
