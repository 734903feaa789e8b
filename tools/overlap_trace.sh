#!/bin/bash
# Kernel-trace timelines of the replayed step with 1 and 2 batch lanes: how much of the time do kernels of different streams really overlap?
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/ov
mkdir -p $O
for L in 1 2; do
  timeout 300 rocprofv3 --kernel-trace -f csv -d $O/l$L -o t -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-breakdown --lanes $L > $O/l$L.log 2>&1
  cp $(find $O/l$L -name "*kernel_trace.csv" | head -1) $O/trace_l$L.csv
  rm -rf $O/l$L
done
ls -la $O; head -2 $O/trace_l1.csv
