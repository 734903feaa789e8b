#!/bin/bash
# round 6, GPU call 11: the graded ResBlock with the one-fragment strip instance (bench.py with its breakdown), the launch-width constant under two
# batch lanes, mmd_aconv (restricted to Cin <= 256) on / off.
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export O=gpurun_out/c11
mkdir -p $O
timeout 600 python bench.py --no-cpu-baseline > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log > $O/full_default.json
MMD_STRIP_K128_RF1=1 timeout 600 python bench.py --no-cpu-baseline > $O/bench_rf1.log 2>&1; tail -1 $O/bench_rf1.log > $O/full_rf1.json
python - <<'PY'
import json, os
for n in ("default", "rf1"):
    d = json.load(open(os.environ["O"] + f"/full_{n}.json"))
    g = d.get("graded", {})
    print(n, "ms_per_step", round(d["ms_per_step"], 3), "resblock", g.get("video_resblock_ds1_128to128", {}).get("ms"), g.get("video_resblock_ds1_128to128", {}).get("frac_of_8TBs"),
          "xattn", g.get("rs_cross_attention_ds2", {}).get("attn_kernels_ms"), g.get("rs_cross_attention_ds2", {}).get("frac_of_mfma_peak"), "roofline frac", d["roofline"]["frac"], d["roofline"].get("frac_in_step"))
PY
B="python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-breakdown"
run() { name=$1; shift; env "$@" timeout 400 $B "${EXTRA[@]}" > $O/b_$name.log 2>&1; tail -1 $O/b_$name.log > $O/line_$name.json; }
for rep in 1 2; do
EXTRA=()
run l1_default_$rep X=1
run l1_aconv0_$rep MMD_ACONV=0
EXTRA=(--lanes 2)
run l2_blocks256_$rep X=1
run l2_blocks128_$rep MMD_STRIP_BLOCKS=128 MMD_TCONV_BLOCKS=128
run l2_blocks192_$rep MMD_STRIP_BLOCKS=192 MMD_TCONV_BLOCKS=192
run l2_pipe_$rep MMD_ATTN_PIPE=1
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
