#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/final
mkdir -p $O
export PYTHONFAULTHANDLER=1
timeout 1500 python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1
timeout 400 python bench.py > $O/bench.log 2>&1
tail -4 $O/pytest.log; tail -1 $O/smoke.log; tail -1 $O/bench.log | cut -c1-1200
