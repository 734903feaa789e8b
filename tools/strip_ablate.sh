#!/bin/bash
# Ablation builds of the row-strip GEMM (mmd_gemm.hip, -DSTRIP_ABLATE=n) timed with tools/strip_probe.py: where a strip launch spends its
# time.  1 no weight-fragment reads / MFMAs, 2 no output stores, 3 no weight DMA after chunk 0, 4 prologue + one chunk only.  Outputs of
# the ablated builds are wrong by construction (the probe's equality column says so); only the times matter.
#   gpurun -- 'bash tools/strip_ablate.sh > gpurun_out/strip_ablate.txt'        (~25 s per variant)
cd "$(dirname "$0")/.."
echo "## product build"; python tools/strip_probe.py
for n in 1 2 3 4; do
  bash tools/build_variant.sh strip_abl$n mmd_gemm.hip "-DSTRIP_ABLATE=$n" > /dev/null
  echo "## STRIP_ABLATE=$n"
  MMD_LIB=mm-diffusion_amd/lib/variants/libmmd_strip_abl$n.so timeout 200 python tools/strip_probe.py
done
# a real candidate, not an ablation: both 32-channel sub-tiles of a chunk in flight (four MFMA chains per wave) for K = 128; results must
# stay bitwise equal (the probe's equality column), only the K = 128 rows can move
bash tools/build_variant.sh strip_joint mmd_gemm.hip "-DSTRIP_JOINT" > /dev/null
echo "## STRIP_JOINT"
MMD_LIB=mm-diffusion_amd/lib/variants/libmmd_strip_joint.so timeout 200 python tools/strip_probe.py
