#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc SQ_* passes (each pass its own run, --kernel-trace only) per kernel: mean counter value per dispatch and
the ratios the MI355X guide reads them by (SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycle wave sums; SQ_BUSY_CYCLES and
SQ_VALU_MFMA_BUSY_CYCLES are cycles).   usage: pmc_sq_summary.py <out.txt> <note> <pass_dir> [<pass_dir> ...]"""
import collections
import csv
import glob
import os
import sys


def load(d):
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(set)
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"]
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[k].add(r.get("Dispatch_Id"))
    return tot, cnt


def main():
    out, note, dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
    per = collections.defaultdict(dict)
    calls = {}
    for d in dirs:
        tot, cnt = load(d)
        for k, cs in tot.items():
            n = max(len(cnt[k]), 1)
            calls[k] = n
            for c, v in cs.items():
                per[k][c] = v / n
    keys = sorted(per, key=lambda k: -per[k].get("SQ_WAVE_CYCLES", per[k].get("SQ_BUSY_CYCLES", 0.0)) * calls.get(k, 1))
    with open(out, "w") as f:
        f.write(f"# rocprofv3 --pmc SQ passes (separate runs, --kernel-trace only) of: {note}\n")
        f.write("# per-dispatch means; ratios: wait = SQ_WAIT_ANY / SQ_WAVE_CYCLES (parked at s_waitcnt / barrier), stall = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES (issue stalls),\n")
        f.write("# issue = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES, mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES * 4 SIMDs-per-CU-normalised: see raw values)\n")
        for k in keys[:16]:
            c = per[k]
            wc = c.get("SQ_WAVE_CYCLES", 0.0)
            f.write(f"\n{k[:150]}  (dispatches {calls[k]})\n")
            for name in sorted(c):
                f.write(f"    {name:32s} {c[name]:16.1f}\n")
            if wc:
                f.write("    ratios: " + "  ".join(f"{lab}={c[n] / wc:.3f}" for lab, n in (("wait", "SQ_WAIT_ANY"), ("stall", "SQ_WAIT_INST_ANY"), ("issue", "SQ_ACTIVE_INST_ANY"),
                                                                                        ("valu", "SQ_ACTIVE_INST_VALU"), ("lds_stall", "SQ_WAIT_INST_LDS"), ("lds", "SQ_ACTIVE_INST_LDS")) if n in c) + "\n")
            if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "SQ_BUSY_CYCLES" in c and c["SQ_BUSY_CYCLES"]:
                r = c["SQ_VALU_MFMA_BUSY_CYCLES"] / c["SQ_BUSY_CYCLES"]
                # SQ_BUSY_CYCLES sums the 32 shader engines (kernel duration x 32), SQ_VALU_MFMA_BUSY_CYCLES the 1024 SIMDs: 32 SIMDs per SE
                f.write(f"    SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES = {r:.3f}  ->  MFMA pipe busy {100 * r / 32:.1f} % of SIMD-cycles "
                        f"(kernel duration ~ {c['SQ_BUSY_CYCLES'] / 32:.0f} cycles)\n")


if __name__ == "__main__":
    main()
