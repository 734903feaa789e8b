#!/bin/bash
# Round 3, GPU call 2: fused halo GroupNorm + deep-ring tile: parity tests, micro-benchmarks, bench A/B lines, then the model-level suites.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c2
mkdir -p $O
export PYTHONFAULTHANDLER=1
timeout 400 python -m pytest tests/test_round3_gpu.py -x -q -m gpu -p no:cacheprovider > $O/pytest_round3.txt 2>&1; echo "round3 tests rc=$?"
tail -15 $O/pytest_round3.txt
timeout 300 python tools/ring_bench.py > $O/ring_bench.txt 2>&1; grep -v amdgpu $O/ring_bench.txt
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --breakdown-out $O/breakdown.json > $O/bench.json 2> $O/bench.err
MMD_HALO_GN=0 timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-breakdown > $O/bench_nohalogn.json 2>> $O/bench.err
MMD_GEMM_RING=0 timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-breakdown > $O/bench_noring.json 2>> $O/bench.err
timeout 500 python -m pytest tests/test_ops_gpu.py tests/test_strip_gpu.py tests/test_model_gpu.py tests/test_configs_gpu.py -x -q -m gpu -p no:cacheprovider > $O/pytest_subset.txt 2>&1
tail -4 $O/pytest_subset.txt
python - <<'PY'
import json
for f in ("bench", "bench_nohalogn", "bench_noring"):
    try:
        d = json.loads(open(f"gpurun_out/c2/{f}.json").read().strip().splitlines()[-1])
        print(f, round(d["ms_per_step"], 3), d.get("graded", {}).get("video_resblock_ds1_128to128", {}).get("ms"), d.get("kernel_ms_per_step"))
    except Exception as e:
        print(f, "failed", e)
PY
tail -5 $O/bench.err
