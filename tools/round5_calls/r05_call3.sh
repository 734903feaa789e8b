#!/bin/bash
# round 5, GPU call 3: two hipGraphExecs launched in turn against one (MMD_GRAPH_EXECS); what the producers' accumulator atomics cost
# by themselves (MMD_GN_TAIL=auto MMD_TAIL_PROBE=acc|none: same plan, producers with / without the integer atomics, nobody finalises).
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export O=gpurun_out/c3
mkdir -p $O
export PYTHONFAULTHANDLER=1
timeout 600 python -m pytest tests/test_model_gpu.py tests/test_round3_gpu.py tests/test_configs_gpu.py -x -q -p no:cacheprovider -k "psample or repeat or replay or configs1 or 250 or drift or rows" > $O/pytest_a.txt 2>&1
tail -4 $O/pytest_a.txt
B="python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-breakdown"
run() { name=$1; shift; env "$@" timeout 300 $B > $O/b_$name.log 2>&1; tail -1 $O/b_$name.log > $O/line_$name.json; }
run execs2_1 X=1
run execs1_1 MMD_GRAPH_EXECS=1
run execs3 MMD_GRAPH_EXECS=3
run execs2_2 X=1
run execs1_2 MMD_GRAPH_EXECS=1
for T in 21 23 24; do
  run tailprobe_none_$T MMD_GN_TAIL=auto MMD_GN_TAIL_MAX=$((1 << T)) MMD_TAIL_PROBE=none
  run tailprobe_acc_$T MMD_GN_TAIL=auto MMD_GN_TAIL_MAX=$((1 << T)) MMD_TAIL_PROBE=acc
done
python - <<'PY' > $O/ab_lines.txt
import json, glob, os
for p in sorted(glob.glob(os.environ["O"] + "/line_*.json")):
    try:
        d = json.load(open(p)); print(f"{os.path.basename(p):34s} ms_per_step {d['ms_per_step']:.3f}  value {d['value']:.1f}")
    except Exception as e:
        print(p, "unreadable", e)
PY
cat $O/ab_lines.txt
