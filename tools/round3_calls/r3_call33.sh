#!/bin/bash
# Round 3, GPU call 33: kernel-trace timeline of the replayed step - busy vs idle per queue
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/c33
mkdir -p $O
timeout 300 rocprofv3 --kernel-trace -f csv -d $O/kt -o t -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-breakdown > $O/kt.log 2>&1
T=$(find $O/kt -name "*kernel_trace.csv" | head -1)
head -1 $T
python tools/timeline_gaps.py $T 4 | tee $O/gaps.txt
python - <<PY
import csv
rows=list(csv.DictReader(open("$T")))
import collections
print(len(rows), "rows; columns", list(rows[0].keys()))
PY
rm -rf $O/kt
