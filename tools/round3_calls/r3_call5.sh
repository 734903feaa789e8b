#!/bin/bash
# Round 3, GPU call 5: halo16 with unrolled taps + descriptor DMA (issue-side diet)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c5
mkdir -p $O
export PYTHONFAULTHANDLER=1
timeout 300 python -m pytest tests/test_round3_gpu.py -q -m gpu -p no:cacheprovider > $O/pytest_round3.txt 2>&1; echo "round3 tests rc=$?"; tail -8 $O/pytest_round3.txt
timeout 200 python tools/halo_bench.py > $O/halo_bench.txt 2>&1; grep -v amdgpu $O/halo_bench.txt
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --breakdown-out $O/breakdown.json > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
for f in ("bench",):
    try:
        d = json.loads(open(f"gpurun_out/c5/{f}.json").read().strip().splitlines()[-1])
        print(f, round(d["ms_per_step"], 3), d.get("graded", {}).get("video_resblock_ds1_128to128", {}).get("ms"), d.get("kernel_ms_per_step"))
    except Exception as e:
        print(f, "failed", e)
PY
tail -3 $O/bench.err
