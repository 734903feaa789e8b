#!/bin/bash
# round 4, GPU call 4: vconv with per-patch laundered lane constants (193 VGPRs instead of 256 + spills)
mkdir -p gpurun_out/c4
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_vconv_gpu.py -x -q > gpurun_out/c4/pytest_vconv.txt 2>&1; echo "rc=$?" >> gpurun_out/c4/pytest_vconv.txt
tail -3 gpurun_out/c4/pytest_vconv.txt
timeout 300 python tools/vconv_bench.py > gpurun_out/c4/vconv_bench.txt 2>&1
cat gpurun_out/c4/vconv_bench.txt
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_round3_gpu.py -x -q -k "attn or attention" > gpurun_out/c4/pytest_attn.txt 2>&1; echo "rc=$?" >> gpurun_out/c4/pytest_attn.txt
tail -5 gpurun_out/c4/pytest_attn.txt
timeout 300 python tools/attn_bench.py > gpurun_out/c4/attn_bench.txt 2>&1
cat gpurun_out/c4/attn_bench.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/c4/bench.json 2> gpurun_out/c4/bench.err
python - <<'PY'
import json
try:
    d=json.loads(open("gpurun_out/c4/bench.json").read().strip().splitlines()[-1])
    print("ms_per_step", d["ms_per_step"], d.get("graded"))
except Exception as e:
    print("ERR", e)
PY
