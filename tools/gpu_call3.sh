#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/c3
mkdir -p $O
export PYTHONFAULTHANDLER=1
timeout 600 python -m pytest tests/test_lifetime_gpu.py tests/test_sampling_api_gpu.py -q -m gpu -p no:cacheprovider > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
for L in 1 2 4; do
  timeout 200 python bench.py --no-cpu-baseline --no-breakdown --lanes $L > $O/bench_l$L.json 2> $O/bench_l$L.err
done
MMD_GEMM_HALO=1 timeout 200 python bench.py --no-cpu-baseline --no-breakdown --lanes 2 > $O/bench_l2_halo.json 2> $O/bench_l2_halo.err
timeout 200 python bench.py --no-cpu-baseline --no-breakdown --lanes 2 --batch 8 > $O/bench_l2_b8.json 2> $O/bench_l2_b8.err
timeout 200 python bench.py --no-cpu-baseline --no-breakdown --lanes 1 --batch 8 > $O/bench_l1_b8.json 2> $O/bench_l1_b8.err
tail -4 $O/pytest.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/c3/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d["ms_per_step"],3), d["value"])
    except Exception as e: print(f, "failed", e)
PY
