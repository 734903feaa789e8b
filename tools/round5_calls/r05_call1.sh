#!/bin/bash
# round 5, GPU call 1: the new kernels' tests, the model fixtures (incl. the new full-size forward), same-call A/B bench lines of every round-5
# plan switch, a kernel trace of the default step by launch shape, the effective-clock pass the round-4 review asked for.
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export O=gpurun_out/c1
mkdir -p $O
export PYTHONFAULTHANDLER=1
timeout 900 python -m pytest tests/test_round5_gpu.py tests/test_model_gpu.py tests/test_vconv_gpu.py -x -q -s -p no:cacheprovider > $O/pytest_a.txt 2>&1
tail -5 $O/pytest_a.txt
timeout 600 python -m pytest tests/test_round3_gpu.py tests/test_lifetime_gpu.py -x -q -p no:cacheprovider > $O/pytest_b.txt 2>&1
tail -3 $O/pytest_b.txt
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -p no:cacheprovider -k "gn or norm or head or resample or stats" > $O/pytest_c.txt 2>&1
tail -3 $O/pytest_c.txt
B="python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-breakdown"
for rep in 1 2; do
  timeout 300 $B > $O/b_default_$rep.log 2>&1; tail -1 $O/b_default_$rep.log > $O/line_default_$rep.json
  MMD_UP_LOWRES=0 timeout 300 $B > $O/b_uplow0_$rep.log 2>&1; tail -1 $O/b_uplow0_$rep.log > $O/line_uplow0_$rep.json
  MMD_CROSS_SERIAL=0 timeout 300 $B > $O/b_cross0_$rep.log 2>&1; tail -1 $O/b_cross0_$rep.log > $O/line_cross0_$rep.json
  MMD_HEAD_GEMM=0 timeout 300 $B > $O/b_head0_$rep.log 2>&1; tail -1 $O/b_head0_$rep.log > $O/line_head0_$rep.json
  MMD_UP_LOWRES=0 MMD_CROSS_SERIAL=0 MMD_HEAD_GEMM=0 MMD_RESAMPLE_STATS=0 timeout 300 $B > $O/b_alloff_$rep.log 2>&1; tail -1 $O/b_alloff_$rep.log > $O/line_alloff_$rep.json
done
python - <<'PY' > $O/ab_lines.txt
import json, glob, os
for p in sorted(glob.glob(os.environ.get("O", "gpurun_out/c1") + "/line_*.json")):
    try:
        d = json.load(open(p)); print(f"{os.path.basename(p):28s} ms_per_step {d['ms_per_step']:.3f}  value {d['value']:.1f}")
    except Exception as e:
        print(p, "unreadable", e)
PY
cat $O/ab_lines.txt
BT="python bench.py --steps 24 --warmup 2 --no-cpu-baseline --no-breakdown"
timeout 600 rocprofv3 --kernel-trace -f csv -d $O/kt -o kt -- $BT > $O/kt.log 2>&1
python tools/kt_by_shape.py "$(find $O/kt -name '*kernel_trace.csv' | head -1)" $O/kt_by_shape.txt 20 > /dev/null
python tools/timeline_gaps.py "$(find $O/kt -name '*kernel_trace.csv' | head -1)" 4 > $O/timeline_gaps.txt 2>&1
BS="python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-breakdown"
timeout 500 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -f csv -d $O/pmc_clk -o p -- $BS > $O/pmc_clk.log 2>&1
python tools/clock_summary.py $O/pmc_clk $O/clocks.txt > /dev/null
timeout 600 python bench.py --no-cpu-baseline --breakdown-out $O/breakdown.json > $O/bench_breakdown.log 2>&1; tail -1 $O/bench_breakdown.log > $O/line_breakdown.json
gzip -c "$(find $O/kt -name '*kernel_trace.csv' | head -1)" > $O/kernel_trace.csv.gz
rm -rf $O/kt $O/pmc_clk
head -60 $O/kt_by_shape.txt; head -30 $O/clocks.txt
