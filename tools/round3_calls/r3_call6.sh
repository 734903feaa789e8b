#!/bin/bash
# Round 3, GPU call 6: descriptor DMA addressing in the direct-to-LDS loops (tiles 129 / 132)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c6
mkdir -p $O
export PYTHONFAULTHANDLER=1
timeout 600 python -m pytest tests/test_round3_gpu.py tests/test_ops_gpu.py -q -m gpu -x -p no:cacheprovider > $O/pytest.txt 2>&1; echo "tests rc=$?"; tail -8 $O/pytest.txt
{ echo "## desc"; timeout 200 python tools/ring_bench.py; echo "## MMD_GEMM_DESC=0"; MMD_GEMM_DESC=0 timeout 200 python tools/ring_bench.py; } > $O/ring_bench.txt 2>&1
grep -v amdgpu $O/ring_bench.txt | cut -c1-200
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --breakdown-out $O/breakdown.json > $O/bench.json 2> $O/bench.err
MMD_GEMM_DESC=0 timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-breakdown > $O/bench_nodesc.json 2>> $O/bench.err
python - <<'PY'
import json
for f in ("bench", "bench_nodesc"):
    try:
        d = json.loads(open(f"gpurun_out/c6/{f}.json").read().strip().splitlines()[-1])
        print(f, round(d["ms_per_step"], 3), d.get("graded", {}).get("video_resblock_ds1_128to128", {}).get("ms"), d.get("kernel_ms_per_step"))
    except Exception as e:
        print(f, "failed", e)
PY
tail -3 $O/bench.err
