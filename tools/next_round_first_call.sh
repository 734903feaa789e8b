#!/bin/bash
# First GPU call of the next round, on this branch merged into main: (1) the whole GPU suite with the fuse / unfuse rule (it has only seen
# 81 of the tests), (2) where a strip launch spends its time and whether four accumulator chains help (tools/strip_ablate.sh),
# (3) the three VALU diets of the attention kernel against the product build (tools/attn_variants.sh), (4) a bench line.
#   gpurun --timeout 1200 -- 'bash tools/next_round_first_call.sh'            (~12 min)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/n1
mkdir -p $O
export PYTHONFAULTHANDLER=1
timeout 900 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $O/pytest_full.txt 2>&1; echo "suite rc=$?" | tee $O/rc.txt
timeout 400 bash tools/strip_ablate.sh > $O/strip_ablate.txt 2>&1; echo "strip ablations rc=$?" | tee -a $O/rc.txt
timeout 300 bash tools/attn_variants.sh > $O/attn_variants.txt 2>&1; echo "attention variants rc=$?" | tee -a $O/rc.txt
timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/rc.txt
tail -3 $O/pytest_full.txt
grep -v "amdgpu.ids" $O/strip_ablate.txt | grep "^##\|sum:\|ds2 qkv (spatial\|ds1 res out\|ds1 up out\|ds2 proj"
grep -v "amdgpu.ids" $O/attn_variants.txt
python - $O/bench.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["ms_per_step"], d["kernel_ms_per_step"], d.get("graded"))
PY
