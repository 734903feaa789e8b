#!/bin/bash
# Round 3, GPU call 31: MFMA stem / head convs, timestep embedding on the audio stream
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c31
mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py tests/test_round3_gpu.py -m gpu -x -q > $O/tests.txt 2>&1; tail -8 $O/tests.txt | cut -c1-300
for kv in "MMD_HEAD_MFMA=1" "MMD_HEAD_MFMA=0 MMD_STEM_MFMA=0" "MMD_HEAD_MFMA=1" "MMD_HEAD_MFMA=0 MMD_STEM_MFMA=0"; do
  env $kv timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json; r=json.loads(sys.stdin.read()); k=r['kernel_ms_per_step']; print('$kv', round(r['ms_per_step'],3), {n:v for n,v in k.items() if 'stem' in n or 'head' in n or 'temb' in n or 'linear' in n})"
done | tee $O/bench_ab.txt
