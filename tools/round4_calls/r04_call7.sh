#!/bin/bash
# round 4, GPU call 7: input norm in quarter-slot pieces behind the MFMA groups (vconv, halo16)
mkdir -p gpurun_out/c7
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_vconv_gpu.py -x -q > gpurun_out/c7/pytest_vconv.txt 2>&1; echo "rc=$?" >> gpurun_out/c7/pytest_vconv.txt
tail -3 gpurun_out/c7/pytest_vconv.txt
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_round3_gpu.py -x -q -k "halo or gn_conv or norm" > gpurun_out/c7/pytest_halo.txt 2>&1; echo "rc=$?" >> gpurun_out/c7/pytest_halo.txt
tail -3 gpurun_out/c7/pytest_halo.txt
timeout 300 python tools/vconv_bench.py > gpurun_out/c7/vconv_bench.txt 2>&1
cat gpurun_out/c7/vconv_bench.txt
timeout 300 python tools/halo_bench.py > gpurun_out/c7/halo_bench.txt 2>&1
tail -8 gpurun_out/c7/halo_bench.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/c7/bench.json 2> gpurun_out/c7/bench.err
python - <<'PY'
import json
try:
    d=json.loads(open("gpurun_out/c7/bench.json").read().strip().splitlines()[-1])
    print("ms_per_step", d["ms_per_step"], d.get("graded"))
except Exception as e:
    print("ERR", e)
PY
