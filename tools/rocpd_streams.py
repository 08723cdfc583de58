#!/usr/bin/env python3
"""per-stream phase summary of the last N ms of a rocpd kernel trace: first/last kernel of each named phase per stream
   python tools/rocpd_streams.py results.db [window_ms]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
win = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
rows = db.execute("select name, start, end, stream_id from kernels order by start").fetchall()
tend = max(r[2] for r in rows)
rows = [r for r in rows if r[1] >= tend - win * 1e6]
base = rows[0][1]


def phase(name):
    n = re.sub(r"\(anonymous namespace\)::", "", name)
    if "k_rans_lanes" in n: return "chain"
    if re.search(r"k_rs_|k_r0_|k_gather_win|k_win_|k_seg_round|k_lg_|k_tab_|k_cmp_|k_bwt_|k_scan_", n): return "bwt"
    if re.search(r"k_enc_|k_rle_|k_cls_|k_quasi|k_adapt|k_pairs|k_density|k_order|k_tile_prefix", n): return "enc-pre"
    if re.search(r"k_emit|k_headers|k_out_offsets|k_put_", n): return "enc-post"
    return "other"


ev = {}
for name, s, e, st in rows:
    ph = phase(name)
    key = (st, ph)
    cur = ev.get(key)
    # split into runs: a gap of more than 3 ms between kernels of the same phase on a stream starts a new run
    if cur and s - cur[-1][1] < 3e6:
        cur[-1][1] = max(cur[-1][1], e); cur[-1][2] += e - s; cur[-1][3] += 1
    else:
        ev.setdefault(key, []).append([s, e, e - s, 1])
out = []
for (st, ph), runs in ev.items():
    for s, e, busy, n in runs:
        out.append((s, f"{(s - base) / 1e6:9.3f} -> {(e - base) / 1e6:9.3f} ms  ({(e - s) / 1e6:7.3f}, busy {busy / 1e6:7.3f}, {n:4d} kernels)  stream {st:3d}  {ph}"))
for _, line in sorted(out):
    print(line)
