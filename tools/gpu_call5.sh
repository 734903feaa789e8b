#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/c5
mkdir -p $O
export PYTHONFAULTHANDLER=1
timeout 120 python tools/lanes_probe.py 2 > $O/probe2.log 2>&1; echo "rc=$?" >> $O/probe2.log
timeout 120 python tools/lanes_probe.py 4 > $O/probe4.log 2>&1; echo "rc=$?" >> $O/probe4.log
timeout 300 python -m pytest tests/test_ops_gpu.py tests/test_lifetime_gpu.py -q -m gpu -p no:cacheprovider -k "attention or lanes or lifetime" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
timeout 120 python tools/attn_bench.py > $O/attn_bench.txt 2>&1
for L in 1 2 4; do
  timeout 200 python bench.py --no-cpu-baseline --no-breakdown --lanes $L > $O/bench_l$L.json 2> $O/bench_l$L.err
done
timeout 200 python bench.py --no-cpu-baseline --lanes 1 > $O/bench_l1_breakdown.json 2> $O/bench_l1_breakdown.err
for f in $O/probe2.log $O/probe4.log; do echo "== $f"; grep -v "^  File\|Extension modules" $f | tail -n 8; done
tail -5 $O/pytest.log; cat $O/attn_bench.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/c5/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d["ms_per_step"],3), d["value"], d.get("graded"))
    except Exception as e: print(f, "failed", e)
PY
