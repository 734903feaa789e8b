#!/bin/bash
# round 6, GPU call 13 (timing-only): the row-strip GEMM without its global stores (everything else kept) - how much of a one-workgroup-per-CU launch is store issue?
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export O=gpurun_out/c13
mkdir -p $O
export STRIP_PROBE_SHAPES=0,2,4,8,10,12,14
echo "== product" >> $O/strip_nostore.txt
timeout 300 python tools/strip_probe.py >> $O/strip_nostore.txt 2>&1
echo "== variant nostore (timing only)" >> $O/strip_nostore.txt
MMD_LIB=$PWD/mm-diffusion_amd/lib/variants/libmmd_strip_nostore.so timeout 300 python tools/strip_probe.py >> $O/strip_nostore.txt 2>&1
grep -v amdgpu.ids $O/strip_nostore.txt | cut -c1-200
