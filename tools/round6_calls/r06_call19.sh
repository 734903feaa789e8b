#!/bin/bash
# round 6, GPU call 19: residual requests of the non-pipelined strip instances pinned in front of the MFMAs (the ds1 out conv had lost 4 %);
# backward tests for the attention block order; graded kernels of the current build.
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export O=gpurun_out/c19
mkdir -p $O
BASE=$PWD/mm-diffusion_amd/lib/variants/libmmd_base.so
timeout 1200 python -m pytest tests/test_bwd_gpu.py tests/test_strip_gpu.py tests/test_round6_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
export STRIP_PROBE_SHAPES=3,5,6,7,9,11,13,15
echo "== base (round start)" > $O/strip_probe.txt
MMD_LIB=$BASE timeout 300 python tools/strip_probe.py >> $O/strip_probe.txt 2>&1
echo "== new" >> $O/strip_probe.txt
timeout 300 python tools/strip_probe.py >> $O/strip_probe.txt 2>&1
grep -v amdgpu.ids $O/strip_probe.txt | cut -c1-250
timeout 600 python bench.py --no-cpu-baseline > $O/bench_full.log 2>&1; tail -1 $O/bench_full.log > $O/full.json
python - <<'PY'
import json, os
d = json.load(open(os.environ["O"] + "/full.json"))
g = d.get("graded", {})
print("ms_per_step", round(d["ms_per_step"], 3), "resblock", g.get("video_resblock_ds1_128to128", {}).get("ms"), g.get("video_resblock_ds1_128to128", {}).get("frac_of_8TBs"),
      "xattn", g.get("rs_cross_attention_ds2", {}).get("attn_kernels_ms"), g.get("rs_cross_attention_ds2", {}).get("frac_of_mfma_peak"), "roofline frac", d["roofline"]["frac"], d["roofline"]["kernel"])
print({k: v for k, v in list(d["kernel_ms_per_step"].items())[:12]})
PY
