#!/bin/bash
# Round 3, GPU call 47: which kernels the vendor GEMM picks for the conv layers' shapes (names carry the macro tile / wave tile / split)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/c47
mkdir -p $O
timeout 200 rocprofv3 --kernel-trace --stats -f csv -d $O/kt -o v -- python tools/vendor_gemm_bench.py > $O/run.log 2>&1
S=$(find $O/kt -name "*kernel_stats.csv" | head -1)
python - <<PY
import csv
rows=list(csv.DictReader(open("$S")))
for r in rows[:40]:
    n=r.get("Name","")
    if "Cijk" in n or "gemm" in n.lower() or "MT" in n:
        print(r.get("Calls"), r.get("AverageNs") or r.get("Average"), n[:400])
PY
rm -rf $O/kt
