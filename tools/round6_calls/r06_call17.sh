#!/bin/bash
# round 6, GPU call 17: blocks per launch of the weight gradient under the XCD-aware order (MMD_WGRAD_BLOCKS sweep, tools/wgrad_bench.py)
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export O=gpurun_out/c17
mkdir -p $O
for b in 256 384 512 768 1024 1536 2048 3072; do
echo "== MMD_WGRAD_BLOCKS=$b" >> $O/sweep.txt
MMD_WGRAD_BLOCKS=$b timeout 300 python tools/wgrad_bench.py 2>&1 | grep -v amdgpu.ids >> $O/sweep.txt
done
echo "== default" >> $O/sweep.txt
timeout 300 python tools/wgrad_bench.py 2>&1 | grep -v amdgpu.ids >> $O/sweep.txt
python - <<'PY'
import os, re, collections
t = collections.OrderedDict(); cur = None
for ln in open(os.environ["O"] + "/sweep.txt"):
    if ln.startswith("=="): cur = ln.strip("= \n"); continue
    m = re.match(r"(.{20})\s+([\d.]+) us", ln)
    if m: t.setdefault(m.group(1).strip(), collections.OrderedDict())[cur] = float(m.group(2))
cols = list(next(iter(t.values())).keys())
print(f"{'shape':26s}" + "".join(f"{c.replace('MMD_WGRAD_BLOCKS=',''):>9s}" for c in cols))
for k, v in t.items():
    print(f"{k:26s}" + "".join(f"{v.get(c, 0):9.1f}" for c in cols))
PY
