#!/bin/bash
# round 6, GPU call 12 (timing-only ablations of the row-strip GEMM's chunk loop): no MFMAs / no epilogue / no LDS fragment reads, on the launches
# of one workgroup per CU with several chunks per workgroup (tools/strip_probe.py shapes 0 .. 4, 12, 14).
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export O=gpurun_out/c12
mkdir -p $O
export STRIP_PROBE_SHAPES=0,1,2,4,10,12,14
echo "== product" >> $O/strip_abl.txt
timeout 300 python tools/strip_probe.py >> $O/strip_abl.txt 2>&1
for v in nomfma noepi nolds; do
  echo "== variant $v (timing only: outputs are wrong)" >> $O/strip_abl.txt
  MMD_LIB=$PWD/mm-diffusion_amd/lib/variants/libmmd_strip_$v.so timeout 300 python tools/strip_probe.py >> $O/strip_abl.txt 2>&1
done
grep -v amdgpu.ids $O/strip_abl.txt | cut -c1-200
