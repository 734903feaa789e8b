#!/bin/bash
# Round 3, GPU call 12: quad-granular GroupNorm records (DPP reduction in the strip epilogue) — parity, then the step time
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c12
mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_strip_gpu.py tests/test_round3_gpu.py tests/test_model_gpu.py -m gpu -x -q > $O/tests.txt 2>&1; tail -15 $O/tests.txt
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --breakdown-out $O/breakdown.json > $O/bench.txt 2>&1; tail -2 $O/bench.txt | cut -c1-900
python - <<'PY'
import json
b = json.load(open("gpurun_out/c12/breakdown.json"))
rows = b.get("kernels") or b.get("rows") or b
try:
    for r in sorted(rows, key=lambda r: -r.get("ms", r.get("total_ms", 0)))[:18]: print(r)
except Exception as e: print(type(b), list(b)[:10] if hasattr(b, "__iter__") else b, e)
PY
