#!/usr/bin/env python3
"""Per-SHAPE kernel durations inside the replayed step from a rocprofv3 --kernel-trace csv (-f csv): dispatches are grouped by
(kernel name, grid, workgroup, LDS) - one launch shape of the plan each - over the last `steps` replayed steps.  The HIP-event breakdown
of bench.py replays the plan eagerly from Python and is host-bound for launches shorter than ~10 us; these are the begin / end
timestamps of the dispatches themselves.
usage: kt_by_shape.py <kernel_trace.csv> <out.txt> [steps]"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("void ", "").replace("bool _Accum", "bf16")
    m = re.match(r"_Z\d+([a-z0-9_]+?)(?:I|Pv|PK|v$)", name)
    return (m.group(1) + "<mangled>" if m else name)[:70]


def main():
    path, out = sys.argv[1], sys.argv[2]
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    rows = list(csv.DictReader(open(path)))
    g = lambda r, k, d="": r.get(k, d)
    ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), g(r, "Queue_Id"), r["Kernel_Name"],
                  "x".join(g(r, f"Grid_Size_{a}", g(r, f"Grid_Size{a}", "?")) for a in "XYZ"), g(r, "Workgroup_Size_X", g(r, "Workgroup_Size", "?")),
                  g(r, "LDS_Block_Size", g(r, "LDS_Block_Size_v", "?"))) for r in rows), key=lambda e: e[0])
    # step boundaries.  With batch lanes a step ends with one fused update per lane and chain, and under the profiler the lanes finish
    # milliseconds apart (round 6: the old 2 ms clustering of the update kernels' end times cut every two-lane step in two - the tables of
    # that round's first profile calls are per LANE-step).  The step's noise draw runs once per step on the origin stream before the lanes
    # fork: the start of the larger of its two launches (torch's normal_ kernel) delimits the steps; without one, the old clustering.
    noise = [e for e in ev if "at::native" in e[3] and ("normal" in e[3] or "distribution" in e[3])]
    if noise:
        big = max(int(e[4].split("x")[0]) if e[4].split("x")[0].isdigit() else 0 for e in noise)
        ends = sorted(e[0] for e in noise if e[4].split("x")[0].isdigit() and int(e[4].split("x")[0]) == big)
    else:
        marks = [e[1] for e in ev if "ddpm_update" in e[3]]
        clusters = []
        for t in sorted(marks):
            if clusters and t - clusters[-1][-1] < 2_000_000:
                clusters[-1].append(t)
            else:
                clusters.append([t])
        ends = [c[-1] for c in clusters]
    if len(ends) < steps + 1:
        steps = max(1, len(ends) - 1)
    t0, t1 = ends[-steps - 1], ends[-1]
    sel = [e for e in ev if t0 <= e[0] and e[0] < t1]
    wall = (t1 - t0) / steps / 1e3
    agg = defaultdict(list)
    for s, e, q, name, grid, wg, lds in sel:
        agg[(short(name), grid, wg, lds)].append((e - s) / 1e3)
    byq = defaultdict(lambda: [0.0, 0])
    for s, e, q, *_ in sel:
        byq[q][0] += (e - s) / 1e3
        byq[q][1] += 1
    with open(out, "w") as f:
        f.write(f"# {path.split('/')[-1]}: last {steps} replayed steps, {len(sel) / steps:.1f} dispatches per step, step wall {wall:.1f} us (under the profiler)\n")
        for q, (busy, n) in sorted(byq.items(), key=lambda kv: -kv[1][0]):
            f.write(f"# queue {q}: {n / steps:.1f} dispatches per step, kernel time {busy / steps:.1f} us per step\n")
        f.write(f"{'per_step':>8} {'us/step':>9} {'avg_us':>8} {'min_us':>8} {'p50_us':>8} {'max_us':>8}  kernel | grid | wg | lds\n")
        tot = 0.0
        for key, d in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            d.sort()
            tot += sum(d)
            f.write(f"{len(d) / steps:8.2f} {sum(d) / steps:9.1f} {sum(d) / len(d):8.2f} {d[0]:8.2f} {d[len(d) // 2]:8.2f} {d[-1]:8.2f}  {key[0]} | {key[1]} | {key[2]} | {key[3]}\n")
        f.write(f"# total kernel time per step {tot / steps:.1f} us\n")
    print(open(out).read()[:6000])


if __name__ == "__main__":
    main()
