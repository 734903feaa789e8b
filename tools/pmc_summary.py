#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (two separate runs, --kernel-trace only) into per-kernel HBM
traffic per launch.  usage: pmc_summary.py <fetch_dir> <write_dir> <out.txt> [<pmc_traffic.json>] [<header note>]

Units / corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): both counters are in KiB; on gfx950 FETCH_SIZE
tallies 128-byte requests as 64 bytes, so hbm_read_bytes = 2 * FETCH_SIZE * 1024 for wide coalesced streams; WRITE_SIZE is
reported as is."""
import collections
import csv
import glob
import json
import os
import sys


def load(d, counter):
    tot, cnt = collections.defaultdict(float), collections.Counter()
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        seen = set()
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] != counter:
                continue
            k = r["Kernel_Name"]
            tot[k] += float(r["Counter_Value"])
            key = (r.get("Dispatch_Id"), k)
            if key not in seen:
                seen.add(key)
                cnt[k] += 1
    return tot, cnt


def label_of(kernel_name):
    """rocprofv3 kernel name -> the label bench.py's breakdown uses for the same launches (bf16 instances only).  rocprofv3 reports
    some instances mangled (_Z16conv_gemm_kernelIDF16bLi128ELi128ELb0EE...) and demangles __bf16 as `bool _Accum` in others."""
    import re
    k = kernel_name
    fp32 = "<float" in k or re.search(r"kernelIf", k) is not None
    m = re.search(r"conv1x1_strip(?:_res)?_kernel(?:<\s*\d+,\s*\d+,\s*\d+,\s*(\d+)|ILi\d+ELi\d+ELi\d+ELi(\d+)E)", k)
    if m:
        return "conv_gemm<bf16,strip>" if (m.group(1) or m.group(2)) == "0" else "gn_conv1x1<bf16,strip>"
    if fp32:
        return None
    if "conv_gemm_glds_kernel" in k:
        return "conv_gemm<bf16,128glds>"
    if "conv_gemm_halo_kernel" in k:
        return "conv_gemm<bf16,128halo>"
    m = re.search(r"conv_gemm_kernelIDF16bLi(\d+)ELi\d+ELb([01])E", k) or re.search(r"conv_gemm_kernel<[^,]+,\s*(\d+),\s*\d+,\s*(true|false|1|0)", k)
    if m:
        return ("gn_conv1x1" if m.group(2) in ("true", "1") else "conv_gemm") + f"<bf16,{m.group(1)}>"
    return None


def main():
    fd, wd, out = sys.argv[1:4]
    jout = sys.argv[4] if len(sys.argv) > 4 else None
    note = sys.argv[5] if len(sys.argv) > 5 else ""
    ft, fc = load(fd, "FETCH_SIZE")
    wt, wc = load(wd, "WRITE_SIZE")
    rows = []
    for k in ft:
        n = fc[k]
        f = ft[k] / n
        w = wt.get(k, 0.0) / max(wc.get(k, 1), 1)
        rows.append((ft[k] + wt.get(k, 0.0), n, f, w, k))
    rows.sort(reverse=True)
    with open(out, "w") as fo:
        fo.write("# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace only)" + (f" of: {note}" if note else "") + "\n")
        fo.write("# counter unit = KiB; gfx950: hbm_read_bytes = 2 * FETCH_SIZE * 1024 (128-B requests tallied at 64 B); WRITE_SIZE as is\n")
        fo.write("  calls  FETCH_KiB/call  read_MB/call(x2)  WRITE_KiB/call  write_MB/call  kernel\n")
        for _, n, f, w, k in rows[:24]:
            fo.write(f"{n:7d} {f:15.1f} {2 * f * 1024 / 1e6:17.2f} {w:15.1f} {w * 1024 / 1e6:14.2f}  {k[:110]}\n")
    if jout:
        # launch-weighted mean per bench.py label (kernel family), so whichever family dominates the step finds its traffic here
        traffic, acc = {}, collections.defaultdict(lambda: [0.0, 0.0])
        for _, n, f, w, k in rows:
            lab = label_of(k)
            if lab:
                acc[lab][0] += n * (2 * f + w) * 1024
                acc[lab][1] += n
        for lab, (num, den) in acc.items():
            if den:
                traffic[lab] = num / den
        traffic["_source"] = f"{os.path.basename(out)}: (2*FETCH_SIZE + WRITE_SIZE) KiB per launch, launch-weighted over the template instances of a kernel family, rocprofv3 --pmc passes"
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        try:                                   # the build the passes were measured on: bench.py only accepts a matching file
            import bench
            traffic["build_id"] = bench.build_id()
        except Exception as e:
            traffic["build_id"] = f"unknown ({e})"
        json.dump(traffic, open(jout, "w"), indent=1)


if __name__ == "__main__":
    main()
