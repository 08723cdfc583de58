#!/usr/bin/env python3
"""per-kernel summary (calls, total, mean, share) of a rocprofv3 --kernel-trace result database (rocpd .db):
   python tools/rocpd_stats.py results.db [divide_by_reps] > profiles/xxx_kernel_stats.txt"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
reps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = db.execute("select name, count(*), sum(duration), min(duration), max(duration) from kernels group by name order by sum(duration) desc").fetchall()
tot = sum(r[2] for r in rows)
span = db.execute("select min(start), max(end) from kernels").fetchone()
print(f"# {sys.argv[1]}: {sum(r[1] for r in rows)} dispatches, kernel time {tot / 1e6:.3f} ms, first start to last end {(span[1] - span[0]) / 1e6:.3f} ms; per-rep = / {reps:g}")
print(f"{'kernel':60s} {'calls/rep':>9s} {'ms/rep':>9s} {'mean us':>9s} {'min us':>8s} {'max us':>8s} {'share':>6s}")
for name, calls, dur, mn, mx in rows:
    short = re.sub(r"\(anonymous namespace\)::", "", name)
    short = re.sub(r"\(.*", "", short)
    print(f"{short[:60]:60s} {calls / reps:9.1f} {dur / 1e6 / reps:9.3f} {dur / calls / 1e3:9.2f} {mn / 1e3:8.2f} {mx / 1e3:8.2f} {100.0 * dur / tot:5.1f}%")
