#!/bin/bash
# Round 3, GPU call 1 (diagnostics at the merged HEAD; every variant library is prebuilt in the build container - the GPU box has no
# product .o files to link against): strip ablations + the four-chain candidate, attention VALU diets, bench line, strip column-split A/B.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c1
mkdir -p $O
export PYTHONFAULTHANDLER=1
V=mm-diffusion_amd/lib/variants
{ echo "## product build"; timeout 200 python tools/strip_probe.py
  for n in abl1 abl2 abl3 abl4 joint; do echo "## strip_$n"; MMD_LIB=$V/libmmd_strip_$n.so timeout 200 python tools/strip_probe.py; done; } > $O/strip_ablate.txt 2>&1
D=/tmp/attn_product_outputs
{ echo "## product build"; ATTN_BENCH_SAVE=$D ATTN_BENCH_IMPLS=2,3 timeout 200 python tools/attn_bench.py
  for n in rowsum bfkv pkfma all; do echo "## attn_$n"; MMD_LIB=$V/libmmd_attn_$n.so ATTN_BENCH_CMP=$D ATTN_BENCH_IMPLS=2 timeout 200 python tools/attn_bench.py; done; } > $O/attn_variants.txt 2>&1
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --breakdown-out $O/breakdown.json > $O/bench.json 2> $O/bench.err
for sb in 256 128; do MMD_STRIP_BLOCKS=$sb timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-breakdown > $O/bench_sb$sb.json 2>> $O/bench.err; done
timeout 200 python -m pytest tests/test_strip_gpu.py tests/test_model_gpu.py -x -q -m gpu -p no:cacheprovider > $O/pytest_subset.txt 2>&1
grep -v "amdgpu.ids" $O/strip_ablate.txt | grep "^##\|sum"
grep -v "amdgpu.ids" $O/attn_variants.txt | cut -c1-200
tail -2 $O/pytest_subset.txt
python - <<'PY'
import json
for f in ("bench", "bench_sb256", "bench_sb128"):
    try:
        d = json.loads(open(f"gpurun_out/c1/{f}.json").read().strip().splitlines()[-1])
        print(f, d["ms_per_step"], d.get("graded"))
    except Exception as e:
        print(f, "failed", e)
PY
