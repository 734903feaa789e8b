#!/bin/bash
# round 6, GPU call 6: mmd_aconv (audio in_layers in one launch) - kernel tests, micro-benchmark against gn_apply + conv_gemm, the model fixtures with it
# in the plan, same-call A/B of the step (MMD_ACONV=0 / 1, MMD_ATTN_PIPE=0 / 1 on the product build).
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export O=gpurun_out/c6
mkdir -p $O
export PYTHONFAULTHANDLER=1
timeout 900 python -m pytest tests/test_round6_gpu.py -x -q -p no:cacheprovider > $O/pytest_round6.txt 2>&1
tail -15 $O/pytest_round6.txt
timeout 600 python tools/aconv_bench.py > $O/aconv_bench.txt 2>&1
grep -v amdgpu.ids $O/aconv_bench.txt
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -p no:cacheprovider > $O/pytest_model.txt 2>&1
tail -5 $O/pytest_model.txt
B="python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-breakdown"
run() { name=$1; shift; env "$@" timeout 300 $B > $O/b_$name.log 2>&1; tail -1 $O/b_$name.log > $O/line_$name.json; }
for rep in 1 2; do
run default_$rep X=1
run aconv0_$rep MMD_ACONV=0
run pipe_$rep MMD_ATTN_PIPE=1
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
