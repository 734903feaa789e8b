#!/usr/bin/env python3
"""Critical-path view of a rocprofv3 --kernel-trace csv of bench.py: per HW queue, busy time vs wall time of one replayed step, the idle
gaps between consecutive dispatches, and the kernels that follow the largest gaps.  usage: timeline_gaps.py trace.csv [steps]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"]) for r in rows), key=lambda e: e[0])
# the replayed steps are the tail of the trace: take the last `steps` occurrences of the step's last kernel as step boundaries
ends = [i for i, e in enumerate(ev) if "ddpm_update" in e[3]]
per = len(ends) // max(1, len(set(e[2] for e in ev if "ddpm_update" in e[3])))
bounds = sorted(set(ev[i][1] for i in ends))
# steps are separated by the LAST ddpm_update of a step (video and audio updates): cluster the update end times
clusters = []
for t in bounds:
    if clusters and t - clusters[-1][-1] < 2_000_000:
        clusters[-1].append(t)
    else:
        clusters.append([t])
marks = [c[-1] for c in clusters]
if len(marks) < 3:
    sys.exit("not enough steps in the trace")
t0, t1 = marks[-3], marks[-2]            # one full replayed step (the one before the last)
step = [e for e in ev if t0 < e[0] <= t1 or t0 < e[1] <= t1]
print(f"step wall {(t1 - t0) / 1e3:.1f} us, {len(step)} dispatches")
byq = defaultdict(list)
for e in step:
    byq[e[2]].append(e)
for q, es in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    es.sort()
    busy = sum(e[1] - e[0] for e in es)
    gaps = [(es[i + 1][0] - es[i][1], es[i][3][:60], es[i + 1][3][:60]) for i in range(len(es) - 1)]
    pos = [g for g in gaps if g[0] > 0]
    print(f"queue {q}: {len(es)} dispatches, busy {busy / 1e3:.1f} us, span {(es[-1][1] - es[0][0]) / 1e3:.1f} us, idle inside span {sum(g[0] for g in pos) / 1e3:.1f} us "
          f"(median gap {sorted(g[0] for g in pos)[len(pos) // 2] / 1e3 if pos else 0:.2f} us)")
    hist = defaultdict(int)
    for g in pos:
        hist[min(int(g[0] / 1000), 20)] += 1
    print("   gap histogram (us: count):", dict(sorted(hist.items())))
    for g in sorted(pos, reverse=True)[:8]:
        print(f"   gap {g[0] / 1e3:7.1f} us  after {g[1]}  before {g[2]}")
# overlap between queues: time with >= 2 kernels in flight
pts = []
for e in step:
    pts.append((e[0], 1))
    pts.append((e[1], -1))
pts.sort()
cur, last, occ = 0, None, defaultdict(int)
for t, d in pts:
    if last is not None:
        occ[min(cur, 3)] += t - last
    cur += d
    last = t
print("time with k kernels in flight (us):", {k: round(v / 1e3, 1) for k, v in sorted(occ.items())})
