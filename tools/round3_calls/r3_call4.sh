#!/bin/bash
# Round 3, GPU call 4: fragment-read hoisting in the tiled main loops (product) against the compiler's own placement (variant nohoist).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c4
mkdir -p $O
export PYTHONFAULTHANDLER=1
V=mm-diffusion_amd/lib/variants
timeout 300 python -m pytest tests/test_round3_gpu.py -x -q -m gpu -p no:cacheprovider > $O/pytest_round3.txt 2>&1; echo "round3 tests rc=$?"; tail -3 $O/pytest_round3.txt
{ echo "## product (hoisted)"; timeout 200 python tools/halo_bench.py; echo "## nohoist"; MMD_LIB=$V/libmmd_nohoist.so timeout 200 python tools/halo_bench.py; } > $O/halo_bench.txt 2>&1
{ echo "## product (hoisted)"; timeout 200 python tools/ring_bench.py; echo "## nohoist"; MMD_LIB=$V/libmmd_nohoist.so timeout 200 python tools/ring_bench.py; } > $O/ring_bench.txt 2>&1
grep -v amdgpu $O/halo_bench.txt; grep -v amdgpu $O/ring_bench.txt | cut -c1-230
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --breakdown-out $O/breakdown.json > $O/bench.json 2> $O/bench.err
MMD_LIB=$V/libmmd_nohoist.so timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-breakdown > $O/bench_nohoist.json 2>> $O/bench.err
MMD_HALO16=0 timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-breakdown > $O/bench_nohalo16.json 2>> $O/bench.err
python - <<'PY'
import json
for f in ("bench", "bench_nohoist", "bench_nohalo16"):
    try:
        d = json.loads(open(f"gpurun_out/c4/{f}.json").read().strip().splitlines()[-1])
        print(f, round(d["ms_per_step"], 3), d.get("graded", {}).get("video_resblock_ds1_128to128", {}).get("ms"), d.get("kernel_ms_per_step"))
    except Exception as e:
        print(f, "failed", e)
PY
tail -3 $O/bench.err
