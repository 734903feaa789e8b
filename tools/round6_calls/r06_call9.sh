#!/bin/bash
# round 6, GPU call 9: the whole GPU suite at the build without the tail mode / with mmd_aconv and the new full-size DPM-Solver++ fixture; batch lanes
# re-measured on today's kernels; which L2 -> fabric write requests make vconv2d1d's WRITE_SIZE 1.41x its output (review item 3c).
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export O=gpurun_out/c9
mkdir -p $O
export PYTHONFAULTHANDLER=1
timeout 2400 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $O/pytest_full.txt 2>&1
tail -4 $O/pytest_full.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 600 python -m pytest tests/test_configs_gpu.py -x -q -s -p no:cacheprovider -k "config4_dpm_solver_pp_full_size" > $O/pytest_cfg4.txt 2>&1; grep "configs\[4\]\|passed\|failed" $O/pytest_cfg4.txt
B="python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-breakdown"
run() { name=$1; shift; env "$@" timeout 400 $B "${EXTRA[@]}" > $O/b_$name.log 2>&1; tail -1 $O/b_$name.log > $O/line_$name.json; }
EXTRA=()
run default_1 X=1
EXTRA=(--lanes 2)
run lanes2_1 X=1
EXTRA=()
run default_2 X=1
EXTRA=(--lanes 2)
run lanes2_2 X=1
EXTRA=()
python - <<'PY' > $O/ab_lines.txt
import json, glob, os
for p in sorted(glob.glob(os.environ["O"] + "/line_*.json")):
    try:
        d = json.load(open(p)); print(f"{os.path.basename(p):34s} ms_per_step {d['ms_per_step']:.3f}  value {d['value']:.1f}")
    except Exception as e:
        print(p, "unreadable", e, open(p.replace('line_', 'b_').replace('.json', '.log')).read()[-300:])
PY
cat $O/ab_lines.txt
rocprofv3 -L 2>/dev/null | grep -o "TCC_[A-Z0-9_]*" | sort -u > $O/tcc_counters.txt; wc -l $O/tcc_counters.txt
V="python tools/vconv_bench.py"
for set in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_EA0_WR_UNCACHED_32B_sum TCC_EA0_WRREQ_DRAM_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_NORMAL_WRITEBACK_sum TCC_ALL_TC_OP_WB_WRITEBACK_sum" "WRITE_SIZE" "FETCH_SIZE"; do
  tag=$(echo $set | tr ' ' '+')
  timeout 300 rocprofv3 --pmc $set --kernel-trace -f csv -d $O/pmc_$tag -o p -- $V > $O/pmc_$tag.log 2>&1
done
python - <<'PY' > $O/vconv_tcc.txt
import csv, glob, os, collections
O = os.environ["O"]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "vconv2d1d_kernel" in k or "gn_apply" in k or "halo16" in k:
            acc[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"    {c:36s} n={len(v):4d} mean {sum(v)/len(v):14.1f}")
PY
cat $O/vconv_tcc.txt
rm -rf $O/pmc_*/
