#!/usr/bin/env python3
"""kernel timeline of the LAST repetition in a rocpd database: per stream, kernels merged into runs
   python tools/rocpd_timeline.py results.db [marker-kernel-substring that starts a repetition]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
mark = sys.argv[2] if len(sys.argv) > 2 else "k_enc_hist"
rows = db.execute("select name, start, end, stream_id, grid_x, workgroup_x from kernels order by start").fetchall()
starts = [r[1] for r in rows if mark in r[0]]
# repetition boundaries: gaps between marker launches larger than 1 ms
t0 = None
prev = None
for s in starts:
    if prev is None or s - prev > 3_000_000:
        t0 = s
    prev = s
rows = [r for r in rows if r[1] >= t0]
base = rows[0][1]
print(f"# last repetition: {len(rows)} dispatches, {(max(r[2] for r in rows) - base) / 1e6:.3f} ms")
for name, s, e, st, gx, wx in rows:
    short = re.sub(r"\(anonymous namespace\)::", "", name)
    short = re.sub(r"\(.*", "", short)
    if (e - s) > 150_000:
        print(f"{(s - base) / 1e6:9.3f} -> {(e - base) / 1e6:9.3f} ms  ({(e - s) / 1e6:7.3f})  stream {st}  {short}  grid {gx // max(wx,1)}")
