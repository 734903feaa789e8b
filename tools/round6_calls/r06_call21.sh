#!/bin/bash
# round 6, GPU call 21: dropout mask from one bernoulli_ kernel instead of rand + compare + copy - training step A/B (MMD_AB_DROPOUT_OLD: a
# switch that exists for this call only), and the keep rate of the new mask.
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export O=gpurun_out/c21
mkdir -p $O
timeout 900 python -m pytest tests/test_train_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
python - <<'PY' 2>&1 | grep -v amdgpu.ids
import sys, os
sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "mm-diffusion_amd"))
import torch
from mm_diffusion import train_ops as T
x = torch.randn(4096, 2048, device="cuda").to(torch.bfloat16).requires_grad_(True)
y = T.DropoutFn.apply(x, 0.1)
keep = (y != 0).float().mean().item()
print("keep rate", keep, "scale check", float((y.float().abs().sum() / x.float().abs().sum())))
y.sum().backward()
print("grad keep", (x.grad != 0).float().mean().item(), "grad value", float(x.grad.max()))
PY
B="python bench.py --mode train --batch 8 --no-cpu-baseline --no-breakdown"
run() { name=$1; shift; env "$@" timeout 600 $B > $O/b_$name.log 2>&1; tail -1 $O/b_$name.log > $O/line_$name.json; }
for rep in 1 2; do
run old_$rep MMD_AB_DROPOUT_OLD=1
run new_$rep X=1
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
