#!/usr/bin/env python3
"""coarse timeline of the last timed compress step of bench.py from a rocprofv3 --kernel-trace csv:
per queue, when the suffix sort / BWT / entropy phases of each block start and end"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
import re
def nm(r):
    m = re.search(r"(k_\w+)", r["Kernel_Name"])
    return m.group(1) if m else r["Kernel_Name"][:20]
# steps start with k_init_keys on two queues; find all k_init_keys
inits = [i for i, r in enumerate(rows) if "k_init_keys" in r["Kernel_Name"]]
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 4      # warmup + steps
blocks_per_step = 2
start = inits[(nsteps - 1) * blocks_per_step]
t0 = int(rows[start]["Start_Timestamp"])
end_idx = inits[nsteps * blocks_per_step] if len(inits) > nsteps * blocks_per_step else len(rows)
ev = {}
for r in rows[start:end_idx]:
    q = r["Queue_Id"]; n = nm(r)
    s = (int(r["Start_Timestamp"]) - t0) / 1e6; e = (int(r["End_Timestamp"]) - t0) / 1e6
    for key, pred in (("sa", lambda n: n in ("k_init_keys",)), ("gather", lambda n: n == "k_bwt_gather"), ("density", lambda n: n == "k_density"),
                      ("rans", lambda n: n == "k_rans_lanes"), ("put", lambda n: n == "k_put_payload")):
        if pred(n):
            ev.setdefault((q, key), []).append((round(s, 2), round(e, 2), r["Grid_Size_X"]))
for k in sorted(ev):
    print(k, ev[k])
