#!/bin/bash
# First GPU run of conv_gemm tile 131 (row-strip 1x1 GEMM): parity tests, micro-benchmark against the paths it replaces, A/B of the
# whole denoising step.  Everything under `timeout`; summaries only under gpurun_out/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/s1
mkdir -p $O
export MMD_GEMM_STRIP=pin
timeout 400 python -m pytest tests/test_strip_gpu.py -q -s -p no:cacheprovider > $O/pytest_strip.log 2>&1
echo "strip tests rc=$?" | tee $O/rc.txt
timeout 200 python tools/strip_probe.py > $O/probe.log 2>&1
echo "probe rc=$?" | tee -a $O/rc.txt
for mode in 0 pin 1x1 gn; do
  MMD_GEMM_STRIP=$mode timeout 200 python bench.py --steps 20 --warmup 3 > $O/bench_$mode.json 2> $O/bench_$mode.err
  echo "bench $mode rc=$?" | tee -a $O/rc.txt
done
if grep -q "passed" $O/pytest_strip.log && ! grep -q "failed" $O/pytest_strip.log; then
  timeout 400 python -m pytest tests/test_model_gpu.py -q -x -p no:cacheprovider > $O/pytest_model.log 2>&1
  echo "model tests (strip on) rc=$?" | tee -a $O/rc.txt
fi
tail -n 40 $O/pytest_strip.log
cat $O/probe.log
for mode in 0 pin 1x1 gn; do python - $O/bench_$mode.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k = d.get("kernel_ms_per_step", {})
    print(sys.argv[1], d["ms_per_step"], {a: round(b, 3) for a, b in k.items() if "gemm" in a or "gn_" in a or "conv1x1" in a}, d.get("graded"))
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
done
tail -n 5 $O/pytest_model.log 2>/dev/null
