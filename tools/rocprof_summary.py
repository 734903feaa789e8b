#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace results.db (rocpd sqlite) into a small text table for profiles/ and, with a third argument,
into the machine-readable profiles/kernel_stats.json that bench.py reads for `roofline.avg_launch_ms_in_step` (stamped with the build
id of the kernel sources + launch plan and with the profiled command, so a stale file is refused).
usage: rocprof_summary.py <results.db> <out.txt> [<out.json> <profiled command>]"""
import json
import os
import sqlite3
import sys

db, out = sys.argv[1], sys.argv[2]
con = sqlite3.connect(db)
rows = con.execute("select name, total_calls, total_duration, average, percentage from top_kernels order by total_duration desc").fetchall()
tot = sum(r[2] for r in rows)
with open(out, "w") as f:
    f.write(f"# rocprofv3 --kernel-trace --stats summary ({db.split('/')[-1]}); durations in us\n")
    if len(sys.argv) > 4:
        f.write(f"# command: {sys.argv[4]}\n")
    f.write(f"# total kernel time {tot/1e3:.3f} ms over {sum(r[1] for r in rows)} dispatches\n")
    f.write(f"{'calls':>8} {'total_us':>14} {'avg_us':>12} {'pct':>7}  name\n")
    for name, calls, total, avg, pct in rows:
        f.write(f"{calls:>8} {total:>14.0f} {avg:>12.1f} {pct:>7.2f}  {name[:160]}\n")
if len(sys.argv) > 3:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    with open(sys.argv[3], "w") as f:
        json.dump({"build_id": bench.build_id(), "command": sys.argv[4] if len(sys.argv) > 4 else "", "total_kernel_ms": tot / 1e3,
                   "kernels": {name: {"calls": calls, "total_us": total, "avg_us": avg} for name, calls, total, avg, _ in rows}}, f, indent=1)
print(open(out).read()[:3000])
