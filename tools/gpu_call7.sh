#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/c7
mkdir -p $O
export PYTHONFAULTHANDLER=1
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -q -m gpu -p no:cacheprovider -x > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
ATTN_BENCH_SHAPES=0,1,2 timeout 120 python tools/attn_bench.py > $O/attn_bench.txt 2>&1
MMD_GN_EPILOGUE=0 timeout 200 python bench.py --no-cpu-baseline --no-breakdown > $O/bench_noepi.json 2> $O/bench_noepi.err
timeout 200 python bench.py --no-cpu-baseline --breakdown-out $O/breakdown.json > $O/bench_epi.json 2> $O/bench_epi.err
tail -15 $O/pytest.log; cat $O/attn_bench.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/c7/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d["ms_per_step"],3), d["value"], d.get("graded")); print(d.get("kernel_ms_per_step"))
    except Exception as e: print(f, "failed", e)
PY
