#!/bin/bash
# round 6, GPU call 5: attn_pipe_kernel at three waves per SIMD (launch bound 3: 168 registers, nine lane constants spilled around the loop)
# against two; each measurement in its own process, impl 5 only (call 4: the second kernel timed in a process runs ~12 % slower).
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export O=gpurun_out/c5
mkdir -p $O
export ATTN_BENCH_SHAPES=0,1,2,4,5
for rep in 1 2; do
for v in w2d2 w3d2; do
  echo "== variant $v (rep $rep)" >> $O/occ.txt
  MMD_LIB=$PWD/mm-diffusion_amd/lib/variants/libmmd_$v.so ATTN_BENCH_IMPLS=5 timeout 300 python tools/attn_bench.py >> $O/occ.txt 2>&1
done
echo "== product impl 4 (rep $rep)" >> $O/occ.txt
ATTN_BENCH_IMPLS=4 timeout 300 python tools/attn_bench.py >> $O/occ.txt 2>&1
done
grep -v amdgpu.ids $O/occ.txt | sed 's/| dma-exact.*//' > $O/occ_clean.txt
cat $O/occ_clean.txt
B="python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-breakdown"
run() { name=$1; shift; env "$@" timeout 300 $B > $O/b_$name.log 2>&1; tail -1 $O/b_$name.log > $O/line_$name.json; }
run default_1 X=1
run pipe_w2_1 MMD_ATTN_PIPE=1 MMD_LIB=$PWD/mm-diffusion_amd/lib/variants/libmmd_w2d2.so
run pipe_w3_1 MMD_ATTN_PIPE=1 MMD_LIB=$PWD/mm-diffusion_amd/lib/variants/libmmd_w3d2.so
run default_2 X=1
run pipe_w2_2 MMD_ATTN_PIPE=1 MMD_LIB=$PWD/mm-diffusion_amd/lib/variants/libmmd_w2d2.so
run pipe_w3_2 MMD_ATTN_PIPE=1 MMD_LIB=$PWD/mm-diffusion_amd/lib/variants/libmmd_w3d2.so
python - <<'PY' > $O/ab_lines.txt
import json, glob, os
for p in sorted(glob.glob(os.environ["O"] + "/line_*.json")):
    try:
        d = json.load(open(p)); print(f"{os.path.basename(p):34s} ms_per_step {d['ms_per_step']:.3f}  value {d['value']:.1f}")
    except Exception as e:
        print(p, "unreadable", e)
PY
cat $O/ab_lines.txt
