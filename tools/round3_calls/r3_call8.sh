#!/bin/bash
# Round 3, GPU call 8: short-slice GroupNorm, selective tails, sturdier autotune: tests + bench A/B
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c8
mkdir -p $O
export PYTHONFAULTHANDLER=1
timeout 600 python -m pytest tests/test_round3_gpu.py tests/test_model_gpu.py -q -m gpu -x -p no:cacheprovider > $O/pytest.txt 2>&1; echo "tests rc=$?"; tail -8 $O/pytest.txt
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --breakdown-out $O/breakdown.json > $O/bench.json 2> $O/bench.err
MMD_GN_TAIL=0 timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-breakdown > $O/bench_notail.json 2>> $O/bench.err
MMD_GN_SMALL=0 timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-breakdown > $O/bench_nosmall.json 2>> $O/bench.err
MMD_GN_TAIL_MAX=8388608 timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-breakdown > $O/bench_tail8m.json 2>> $O/bench.err
python - <<'PY'
import json
for f in ("bench", "bench_notail", "bench_nosmall", "bench_tail8m"):
    try:
        d = json.loads(open(f"gpurun_out/c8/{f}.json").read().strip().splitlines()[-1])
        print(f, round(d["ms_per_step"], 3), d.get("graded", {}).get("video_resblock_ds1_128to128", {}).get("ms"), d.get("kernel_ms_per_step"))
    except Exception as e:
        print(f, "failed", e)
PY
tail -5 $O/bench.err
