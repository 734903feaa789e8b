#!/bin/bash
# Round 3, GPU call 3: the 16 x 16-patch halo tile (133): parity, micro-benchmark against tiles 129 / 130, bench A/B.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c3
mkdir -p $O
export PYTHONFAULTHANDLER=1
timeout 400 python -m pytest tests/test_round3_gpu.py -x -q -m gpu -p no:cacheprovider > $O/pytest_round3.txt 2>&1; echo "round3 tests rc=$?"
tail -12 $O/pytest_round3.txt
timeout 300 python tools/halo_bench.py > $O/halo_bench.txt 2>&1; grep -v amdgpu $O/halo_bench.txt
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --breakdown-out $O/breakdown.json > $O/bench.json 2> $O/bench.err
MMD_HALO16=0 timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-breakdown > $O/bench_nohalo16.json 2>> $O/bench.err
python - <<'PY'
import json
for f in ("bench", "bench_nohalo16"):
    try:
        d = json.loads(open(f"gpurun_out/c3/{f}.json").read().strip().splitlines()[-1])
        print(f, round(d["ms_per_step"], 3), d.get("graded", {}).get("video_resblock_ds1_128to128", {}).get("ms"), d.get("kernel_ms_per_step"))
    except Exception as e:
        print(f, "failed", e)
PY
tail -5 $O/bench.err
