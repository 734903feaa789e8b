#!/bin/bash
# round 6, GPU call 33: the per-shape kernel trace again with the step delimited by the noise draw (the two-lane steps were cut in two)
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c33
mkdir -p $O
BC="python bench.py --steps 24 --warmup 2 --no-cpu-baseline --no-breakdown"
timeout 600 rocprofv3 --kernel-trace -f csv -d $O/ktc -o kt -- $BC > $O/ktc.log 2>&1
CSV="$(find $O/ktc -name '*kernel_trace.csv' | head -1)"
python tools/kt_by_shape.py "$CSV" $O/r06_kernel_trace_by_shape.txt 20 > /dev/null
python tools/timeline_gaps.py "$CSV" 4 > $O/timeline_gaps.txt 2>&1
grep -o 'at::native[^"]*' "$CSV" | sort | uniq -c | sort -rn | head -5 > $O/native_names.txt
rm -rf $O/ktc
head -8 $O/r06_kernel_trace_by_shape.txt; cat $O/native_names.txt; head -40 $O/timeline_gaps.txt
