#!/usr/bin/env python3
"""how busy the GPU is in the last `window_ms` of a rocpd kernel trace: time with >= 1 kernel running, with only
"narrow" kernels (grid < 1024 workgroups) running, mean number of kernels in flight, and kernel time by phase
   python tools/rocpd_busy.py results.db [window_ms]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
win = float(sys.argv[2]) if len(sys.argv) > 2 else 300.0
cols = [r[1] for r in db.execute("pragma table_info(kernels)").fetchall()]
gx = "grid_x" if "grid_x" in cols else ("grid_size_x" if "grid_size_x" in cols else None)
wx = "workgroup_x" if "workgroup_x" in cols else ("workgroup_size_x" if "workgroup_size_x" in cols else None)
sel = "name, start, end" + (f", {gx}, grid_y, grid_z, {wx}" if gx and wx and gx == "grid_x" else (f", {gx}, grid_size_y, grid_size_z, {wx}" if gx and wx else ""))
rows = db.execute(f"select {sel} from kernels order by start").fetchall()
tend = max(r[2] for r in rows)
rows = [r for r in rows if r[1] >= tend - win * 1e6]
t0 = rows[0][1]
ev = []
for r in rows:
    wide = True
    if len(r) > 3 and r[6]:
        threads = r[3] * r[4] * r[5]
        wide = threads >= 1024 * 64 * 4            # at least 4 waves per SIMD's worth of work items
    ev.append((r[1], 1, wide))
    ev.append((r[2], -1, wide))
ev.sort()
n = nw = 0
last = t0
busy = narrow_only = area = 0
for t, d, wide in ev:
    dt = t - last
    if n > 0:
        busy += dt
        area += dt * n
        if nw == 0:
            narrow_only += dt
    n += d
    nw += d if wide else 0
    last = t
span = tend - t0
print(f"window {span / 1e6:.1f} ms: >=1 kernel running {100 * busy / span:.1f}%, only narrow kernels (< 4 waves/SIMD) {100 * narrow_only / span:.1f}%, "
      f"idle {100 * (span - busy) / span:.1f}%, mean kernels in flight {area / span:.2f}")
tot = {}
for r in rows:
    nm = re.sub(r"\(anonymous namespace\)::|void |<.*", "", r[0])
    tot[nm] = tot.get(nm, 0) + (r[2] - r[1])
for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:25]:
    print(f"  {k:28s} {v / 1e6:9.3f} ms  ({100 * v / span:5.1f}% of the window)")
