#!/bin/bash
# round 6, GPU call 1: the hand-written pipelined attention kernel (mmd_attn_fwd impl 5) - bitwise tests against impl 2, the micro-benchmark
# against impl 4 (DMA-staged, compiler-scheduled) on the model's attention shapes, same-call A/B bench lines with it as the default.
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export O=gpurun_out/c1
mkdir -p $O
export PYTHONFAULTHANDLER=1
timeout 900 python -m pytest tests/test_round6_gpu.py -x -q -p no:cacheprovider > $O/pytest_attn_pipe.txt 2>&1
tail -15 $O/pytest_attn_pipe.txt
timeout 600 python tools/attn_bench.py > $O/attn_bench.txt 2>&1
cat $O/attn_bench.txt
B="python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-breakdown"
run() { name=$1; shift; env "$@" timeout 300 $B > $O/b_$name.log 2>&1; tail -1 $O/b_$name.log > $O/line_$name.json; }
run default_1 X=1
run pipe_1 MMD_ATTN_PIPE=1
run default_2 X=1
run pipe_2 MMD_ATTN_PIPE=1
python - <<'PY' > $O/ab_lines.txt
import json, glob, os
for p in sorted(glob.glob(os.environ["O"] + "/line_*.json")):
    try:
        d = json.load(open(p)); print(f"{os.path.basename(p):34s} ms_per_step {d['ms_per_step']:.3f}  value {d['value']:.1f}")
    except Exception as e:
        print(p, "unreadable", e)
PY
cat $O/ab_lines.txt
