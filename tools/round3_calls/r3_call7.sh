#!/bin/bash
# Round 3, GPU call 7: in-launch GroupNorm tails (statistics + affine inside the producer launches)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c7
mkdir -p $O
export PYTHONFAULTHANDLER=1
timeout 600 python -m pytest tests/test_round3_gpu.py -q -m gpu -x -p no:cacheprovider > $O/pytest_round3.txt 2>&1; echo "round3 tests rc=$?"; tail -25 $O/pytest_round3.txt
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_strip_gpu.py tests/test_ops_gpu.py -q -m gpu -x -p no:cacheprovider > $O/pytest_model.txt 2>&1; echo "model tests rc=$?"; tail -8 $O/pytest_model.txt
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --breakdown-out $O/breakdown.json > $O/bench.json 2> $O/bench.err
MMD_GN_TAIL=0 timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-breakdown > $O/bench_notail.json 2>> $O/bench.err
python - <<'PY'
import json
for f in ("bench", "bench_notail"):
    try:
        d = json.loads(open(f"gpurun_out/c7/{f}.json").read().strip().splitlines()[-1])
        print(f, round(d["ms_per_step"], 3), d.get("graded", {}).get("video_resblock_ds1_128to128", {}).get("ms"), d.get("kernel_ms_per_step"))
    except Exception as e:
        print(f, "failed", e)
PY
tail -5 $O/bench.err
