#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace results.db (rocpd sqlite) into a small text table for profiles/."""
import sqlite3
import sys

db, out = sys.argv[1], sys.argv[2]
con = sqlite3.connect(db)
rows = con.execute("select name, total_calls, total_duration, average, percentage from top_kernels order by total_duration desc").fetchall()
tot = sum(r[2] for r in rows)
with open(out, "w") as f:
    f.write(f"# rocprofv3 --kernel-trace --stats summary ({db.split('/')[-1]}); durations in us\n")
    f.write(f"# total kernel time {tot/1e3:.3f} ms over {sum(r[1] for r in rows)} dispatches\n")
    f.write(f"{'calls':>8} {'total_us':>14} {'avg_us':>12} {'pct':>7}  name\n")
    for name, calls, total, avg, pct in rows:
        f.write(f"{calls:>8} {total:>14.0f} {avg:>12.1f} {pct:>7.2f}  {name[:160]}\n")
print(open(out).read()[:3000])
