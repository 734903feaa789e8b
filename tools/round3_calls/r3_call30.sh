#!/bin/bash
# Round 3, GPU call 30: the build without packed fp32 in the norm kernels - replay stress, tests, and what a library-wide switch would cost
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c30
mkdir -p $O
{ timeout 100 python tools/determinism_stress.py mid 3000
  MMD_GEMM_STRIP=0 timeout 100 python tools/determinism_stress.py mid 1500
  MMD_GEMM_RING=0 timeout 100 python tools/determinism_stress.py mid 1500
  timeout 200 python tools/determinism_stress.py full 400 noise; } 2>&1 | grep -v amdgpu > $O/stress.txt
cut -c1-300 $O/stress.txt
timeout 900 python -m pytest tests/test_round3_gpu.py tests/test_ops_gpu.py tests/test_multirank_gpu.py -m gpu -q > $O/tests.txt 2>&1; tail -5 $O/tests.txt | cut -c1-300
for n in product nopk product nopk; do
  L=mm-diffusion_amd/lib/libmmd.so; [ $n = nopk ] && L=mm-diffusion_amd/lib/variants/libmmd_nopk.so
  MMD_LIB=$L timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-breakdown 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$n', r['ms_per_step'])"
done | tee $O/bench_ab.txt
