#!/bin/bash
# round 4, GPU call 1: fused VideoConv parity + timing, then the step
mkdir -p gpurun_out/c1
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_vconv_gpu.py -x -q > gpurun_out/c1/pytest_vconv.txt 2>&1
echo "pytest rc=$?" >> gpurun_out/c1/pytest_vconv.txt
tail -15 gpurun_out/c1/pytest_vconv.txt
timeout 300 python tools/vconv_bench.py > gpurun_out/c1/vconv_bench.txt 2>&1
cat gpurun_out/c1/vconv_bench.txt
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/c1/bench_fused.json 2> gpurun_out/c1/bench_fused.err
tail -c 1500 gpurun_out/c1/bench_fused.json
MMD_VCONV_FUSED=0 timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/c1/bench_unfused.json 2> gpurun_out/c1/bench_unfused.err
python - <<'PY'
import json
for n in ("fused","unfused"):
    try:
        d=json.loads(open(f"gpurun_out/c1/bench_{n}.json").read().strip().splitlines()[-1])
        print(n, d["ms_per_step"], d.get("graded",{}).get("video_resblock_ds1_128to128"))
    except Exception as e:
        print(n, "ERR", e)
PY
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_configs_gpu.py -x -q > gpurun_out/c1/pytest_model.txt 2>&1
tail -5 gpurun_out/c1/pytest_model.txt
