#!/bin/bash
# One gpurun that decides whether conv_gemm tile 130 (halo-tile 3x3 / temporal-k3 main loop) earns a place in the plan:
#   gpurun --timeout 600 -- 'bash tools/try_halo.sh'
# 1) its parity tests (skipped by default), 2) the per-shape microbench (tile 130 beside 64 / 128 / 129), 3) the headline bench with
# and without tile 130 among the autotune candidates.  Outputs under gpurun_out/halo/.
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/halo
mkdir -p $O
MMD_TEST_EXPERIMENTAL=1 timeout 200 python -m pytest tests/test_ops_gpu.py -q -k "halo" 2>&1 | tail -15 > $O/pytest.txt
timeout 200 python tools/gemm_bench.py > $O/gemm_bench.txt 2>&1
timeout 120 python bench.py --no-cpu-baseline --no-breakdown > $O/bench_base.json 2> $O/bench_base.err
MMD_GEMM_HALO=1 timeout 150 python bench.py --no-cpu-baseline --breakdown-out $O/breakdown_halo.json > $O/bench_halo.json 2> $O/bench_halo.err
tail -5 $O/pytest.txt
grep -E "3x3|k3t" $O/gemm_bench.txt
python - <<'PY'
import json
for f in ("bench_base", "bench_halo"):
    try:
        d = json.loads(open(f"gpurun_out/halo/{f}.json").read().strip().splitlines()[-1])
        print(f, d["ms_per_step"], d.get("roofline", {}).get("frac"))
    except Exception as e:
        print(f, "failed:", e)
PY
