#!/usr/bin/env python3
"""Effective shader clock per kernel: GRBM_GUI_ACTIVE (rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -f csv; a pass of its own) divided by
the dispatch's wall time from the same run's kernel trace (MI355X_MICROARCH.md, DVFS give-back: effective clock ~ GRBM_GUI_ACTIVE / wall).
The counter is reported per XCC dimension or summed, depending on the rocprofv3 build: both readings are printed (sum / wall and
sum / 8 / wall); the one that lands in 1.2 - 2.4 GHz is the clock.
usage: clock_summary.py <pmc_dir> <out.txt> [name filter regex]"""
import collections
import csv
import glob
import os
import re
import sys


def main():
    d, out = sys.argv[1], sys.argv[2]
    flt = re.compile(sys.argv[3]) if len(sys.argv) > 3 else None
    cyc = collections.defaultdict(float)
    nrows = collections.Counter()
    name = {}
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] != "GRBM_GUI_ACTIVE":
                continue
            k = r["Dispatch_Id"]
            cyc[k] += float(r["Counter_Value"])
            nrows[k] += 1
            name[k] = r["Kernel_Name"]
    dur = {}
    for path in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3      # us
    agg = collections.defaultdict(lambda: [0.0, 0.0, 0, 0])
    for k, c in cyc.items():
        if k not in dur or dur[k] <= 0 or (flt and not flt.search(name[k])):
            continue
        a = agg[re.sub(r"\(.*$", "", name[k])[:90]]
        a[0] += c
        a[1] += dur[k]
        a[2] += 1
        a[3] = max(a[3], nrows[k])
    with open(out, "w") as f:
        f.write("# effective clock per kernel = GRBM_GUI_ACTIVE / dispatch wall time (rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace, own pass)\n")
        f.write("# rows/dispatch = counter rows per dispatch in the csv (1: already aggregated, 8: one per XCC)\n")
        f.write(f"{'calls':>7} {'avg_us':>9} {'cycles/call':>13} {'GHz(sum)':>9} {'GHz(sum/8)':>11} {'rows':>5}  kernel\n")
        for k, (c, t, n, nr) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            f.write(f"{n:7d} {t / n:9.2f} {c / n:13.0f} {c / t / 1e3:9.3f} {c / 8 / t / 1e3:11.3f} {nr:5d}  {k}\n")
    print(open(out).read()[:5000])


if __name__ == "__main__":
    main()
