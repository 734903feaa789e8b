#!/bin/bash
# round 4, GPU call 3: persistent vconv, halo16 with interleaved DMA issue, attention back on the exact loop
mkdir -p gpurun_out/c3
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_vconv_gpu.py -x -q > gpurun_out/c3/pytest_vconv.txt 2>&1; echo "rc=$?" >> gpurun_out/c3/pytest_vconv.txt
tail -4 gpurun_out/c3/pytest_vconv.txt
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_round3_gpu.py -x -q -k "attn or attention or halo" > gpurun_out/c3/pytest_attn_halo.txt 2>&1; echo "rc=$?" >> gpurun_out/c3/pytest_attn_halo.txt
tail -4 gpurun_out/c3/pytest_attn_halo.txt
timeout 300 python tools/vconv_bench.py > gpurun_out/c3/vconv_bench.txt 2>&1
cat gpurun_out/c3/vconv_bench.txt
timeout 300 python tools/halo_bench.py > gpurun_out/c3/halo_bench.txt 2>&1
tail -12 gpurun_out/c3/halo_bench.txt
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/c3/bench.json 2> gpurun_out/c3/bench.err
python - <<'PY'
import json
try:
    d=json.loads(open("gpurun_out/c3/bench.json").read().strip().splitlines()[-1])
    print("ms_per_step", d["ms_per_step"], d.get("graded"))
except Exception as e:
    print("ERR", e)
PY
timeout 1200 python -m pytest tests/test_model_gpu.py tests/test_configs_gpu.py tests/test_sr_gpu.py -x -q > gpurun_out/c3/pytest_model.txt 2>&1; echo "rc=$?" >> gpurun_out/c3/pytest_model.txt
tail -4 gpurun_out/c3/pytest_model.txt
