#!/bin/bash
# round 5, GPU call 13: how many workgroups a launch of the small levels should spread over.  Call 12: MMD_STRIP_BLOCKS=256 10.80 ms against
# 11.14 ms at the default 448 (192: 10.96, 320: 11.02, 128: 11.40).  Finer sweep, twice; the same knob for the temporal conv and gn_apply;
# the fuse / unfuse rule of the strip GEMM's GroupNorm following the new split (MMD_STRIP_BLOCKS_RULE).
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export O=gpurun_out/c13
mkdir -p $O
B="python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-breakdown"
run() { name=$1; shift; env "$@" timeout 300 $B > $O/b_$name.log 2>&1; tail -1 $O/b_$name.log > $O/line_$name.json; }
run s448_a X=1
run s256_a MMD_STRIP_BLOCKS=256
run s224_a MMD_STRIP_BLOCKS=224
run s288_a MMD_STRIP_BLOCKS=288
run s256_rule256 MMD_STRIP_BLOCKS=256 MMD_STRIP_BLOCKS_RULE=256
run s256_t256 MMD_STRIP_BLOCKS=256 MMD_TCONV_BLOCKS=256
run s256_t384 MMD_STRIP_BLOCKS=256 MMD_TCONV_BLOCKS=384
run s256_g1024 MMD_STRIP_BLOCKS=256 MMD_GN_APPLY_BLOCKS=1024
run s256_g512 MMD_STRIP_BLOCKS=256 MMD_GN_APPLY_BLOCKS=512
run s448_b X=1
run s256_b MMD_STRIP_BLOCKS=256
run s224_b MMD_STRIP_BLOCKS=224
run s288_b MMD_STRIP_BLOCKS=288
run s256_t256_g1024 MMD_STRIP_BLOCKS=256 MMD_TCONV_BLOCKS=256 MMD_GN_APPLY_BLOCKS=1024
python - <<'PY' > $O/ab_lines.txt
import json, glob, os
for p in sorted(glob.glob(os.environ["O"] + "/line_*.json")):
    try:
        d = json.load(open(p)); print(f"{os.path.basename(p):34s} ms_per_step {d['ms_per_step']:.3f}  value {d['value']:.1f}")
    except Exception as e:
        print(p, "unreadable", e)
PY
cat $O/ab_lines.txt
