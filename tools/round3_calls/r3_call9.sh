#!/bin/bash
# Round 3, GPU call 9: DMA-staged attention kernel (impl 4): parity + micro-benchmark + bench A/B
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c9
mkdir -p $O
export PYTHONFAULTHANDLER=1
timeout 600 python -m pytest tests/test_round3_gpu.py -q -m gpu -x -p no:cacheprovider -k "attn_dma" > $O/pytest.txt 2>&1; echo "tests rc=$?"; tail -8 $O/pytest.txt
ATTN_BENCH_IMPLS=2,3,4 timeout 300 python tools/attn_bench.py > $O/attn_bench.txt 2>&1; grep -v amdgpu $O/attn_bench.txt | cut -c1-260
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --breakdown-out $O/breakdown.json > $O/bench.json 2> $O/bench.err
MMD_ATTN_DMA=0 timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-breakdown > $O/bench_nodma.json 2>> $O/bench.err
python - <<'PY'
import json
for f in ("bench", "bench_nodma"):
    try:
        d = json.loads(open(f"gpurun_out/c9/{f}.json").read().strip().splitlines()[-1])
        print(f, round(d["ms_per_step"], 3), d.get("graded", {}).get("rs_cross_attention_ds2"), d.get("kernel_ms_per_step"))
    except Exception as e:
        print(f, "failed", e)
PY
tail -3 $O/bench.err
