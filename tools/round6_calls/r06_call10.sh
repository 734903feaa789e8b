#!/bin/bash
# round 6, GPU call 10: (a) the DMA-staged weight-gradient kernel with the bias gradient inside (every tap count, no colsum launch) - backward tests
# and same-call A/B of the training step (MMD_WGRAD_TR=9 = the round-5 routing); (b) the one-fragment K = 128 row-strip instance (128-row blocks
# for the ds1 ResBlock out conv) - test and A/B; (c) batch lanes 2 / 4 again.
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export O=gpurun_out/c10
mkdir -p $O
export PYTHONFAULTHANDLER=1
timeout 1200 python -m pytest tests/test_bwd_gpu.py tests/test_train_gpu.py tests/test_round6_gpu.py tests/test_strip_gpu.py -x -q -p no:cacheprovider > $O/pytest_a.txt 2>&1
tail -4 $O/pytest_a.txt
timeout 300 python tools/wgrad_bench.py > $O/wgrad_bench_new.txt 2>&1
MMD_WGRAD_TR=9 timeout 300 python tools/wgrad_bench.py > $O/wgrad_bench_old.txt 2>&1
grep -v amdgpu.ids $O/wgrad_bench_new.txt | head -30; echo ---; grep -v amdgpu.ids $O/wgrad_bench_old.txt | head -30
BT="python bench.py --mode train --batch 8 --steps 5 --warmup 2"
for rep in 1 2; do
  timeout 400 $BT > $O/bt_new_$rep.log 2>&1; tail -1 $O/bt_new_$rep.log > $O/line_train_new_$rep.json
  MMD_WGRAD_TR=9 timeout 400 $BT > $O/bt_old_$rep.log 2>&1; tail -1 $O/bt_old_$rep.log > $O/line_train_old_$rep.json
done
B="python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-breakdown"
run() { name=$1; shift; env "$@" timeout 400 $B "${EXTRA[@]}" > $O/b_$name.log 2>&1; tail -1 $O/b_$name.log > $O/line_$name.json; }
for rep in 1 2; do
EXTRA=()
run default_$rep X=1
run rf1_$rep MMD_STRIP_K128_RF1=1
EXTRA=(--lanes 2)
run lanes2_$rep X=1
run lanes2_rf1_$rep MMD_STRIP_K128_RF1=1
EXTRA=(--lanes 4)
run lanes4_$rep X=1
done
python - <<'PY' > $O/ab_lines.txt
import json, glob, os
for p in sorted(glob.glob(os.environ["O"] + "/line_*.json")):
    try:
        d = json.load(open(p)); print(f"{os.path.basename(p):34s} ms_per_step {d['ms_per_step']:.3f}  value {d['value']:.1f}")
    except Exception as e:
        print(p, "unreadable", e, open(p.replace('line_', 'b_').replace('.json', '.log')).read()[-300:] if os.path.exists(p.replace('line_', 'b_').replace('.json', '.log')) else "")
PY
cat $O/ab_lines.txt
