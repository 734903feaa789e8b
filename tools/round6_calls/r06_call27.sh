#!/bin/bash
# round 6, GPU call 27: the fused VideoConv's two-slot weight ring (121.75 KB of LDS) against the three-slot ring (145.75 KB):
# bitwise tests, the kernel alone (one process per arm), the step (alternating, three passes), the graded block
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export O=gpurun_out/c27
mkdir -p $O
timeout 900 python -m pytest tests/test_vconv_gpu.py tests/test_round6_gpu.py -m gpu -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for rep in 1 2; do
for ring in 3 2; do MMD_VCONV_RING=$ring timeout 300 python tools/vconv_bench.py > $O/vconv_ring${ring}_$rep.txt 2>&1; done
done
B="python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-breakdown"
run() { name=$1; shift; env "$@" timeout 400 $B > $O/b_$name.log 2>&1; tail -1 $O/b_$name.log > $O/line_$name.json; }
for rep in 1 2 3; do
run ring3_$rep MMD_VCONV_RING=3
run ring2_$rep MMD_VCONV_RING=2
done
for ring in 3 2; do MMD_VCONV_RING=$ring timeout 500 python bench.py --no-cpu-baseline > $O/full_ring$ring.log 2>&1; tail -1 $O/full_ring$ring.log > $O/line_full_ring$ring.json; done
python - <<'PY' > $O/ab_lines.txt
import json, glob, os
for p in sorted(glob.glob(os.environ["O"] + "/line_*.json")):
    try:
        d = json.load(open(p)); g = d.get("graded", {}).get("video_resblock_ds1_128to128", {}).get("ms")
        print(f"{os.path.basename(p):34s} ms_per_step {d['ms_per_step']:.3f}  value {d['value']:.1f}  graded1 {g}")
    except Exception as e:
        print(p, "unreadable", e)
PY
cat $O/vconv_ring*.txt $O/ab_lines.txt
