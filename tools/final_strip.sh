#!/bin/bash
# Last GPU call of round 2: first run of the full conv_gemm tile 131 feature set (k=3 convs at 128 channels, K = 512, half-record
# statistics), then tools/round_profile.sh at whatever strip mode survives its own parity tests.  If the new features fail, the DEFAULT
# in ops.py is rewritten on this box (pin -> base -> 0) before anything else runs, and gpurun_out/<tag>/strip_mode.txt says so: the same
# one-line edit is then made in the repository, so the profiled tree and the committed tree are the same files (same bench.build_id()).
set -x
TAG=${1:-r02}
cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG
mkdir -p $O profiles
OPS=mm-diffusion_amd/mm_diffusion/ops.py
mode=pin
if ! timeout 400 python -m pytest tests/test_strip_gpu.py -q -s -p no:cacheprovider > $O/pytest_strip_pin.log 2>&1; then
  mode=base
  sed -i 's/os.environ.get("MMD_GEMM_STRIP", "pin")/os.environ.get("MMD_GEMM_STRIP", "base")/' $OPS
  if ! timeout 400 python -m pytest tests/test_strip_gpu.py -q -s -p no:cacheprovider > $O/pytest_strip_base.log 2>&1; then
    mode=0
    sed -i 's/os.environ.get("MMD_GEMM_STRIP", "base")/os.environ.get("MMD_GEMM_STRIP", "0")/' $OPS
  fi
fi
echo $mode > $O/strip_mode.txt
grep -n 'MMD_GEMM_STRIP", "' $OPS >> $O/strip_mode.txt
tail -n 30 $O/pytest_strip_pin.log
timeout 300 python tools/strip_probe.py > profiles/${TAG}_strip_probe.txt 2>&1
MMD_GEMM_STRIP=0 timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_strip_off.log 2>&1
tail -1 $O/bench_strip_off.log > profiles/${TAG}_bench_line_strip_off.json
if [ $mode = pin ]; then
  MMD_GEMM_STRIP=base timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_strip_base.log 2>&1
  tail -1 $O/bench_strip_base.log > profiles/${TAG}_bench_line_strip_base.json
fi
bash tools/round_profile.sh $TAG
cat $O/strip_mode.txt
cat profiles/${TAG}_strip_probe.txt
python - <<'PY'
import json, glob
for f in sorted(glob.glob("profiles/r02_bench_line*.json")):
    try:
        d = json.loads(open(f).read())
        print(f, d.get("ms_per_step"), d.get("value"), (d.get("roofline") or {}).get("kernel"), (d.get("roofline") or {}).get("traffic"), d.get("graded"))
    except Exception as e:
        print(f, "unreadable", e)
PY
