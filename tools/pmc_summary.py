#!/usr/bin/env python3
"""Summarise two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; KiB per dispatch) per kernel.
   python tools/pmc_summary.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE out.json [passes of the profiled command = 2]
gfx950 note (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports 1/2 of the bytes of a wide coalesced 16 B/lane stream;
other access widths are uncalibrated, so both the raw and the x2-corrected read figure are kept."""
import csv
import glob
import json
import re
import sys
from collections import defaultdict


def load(d, name):
    f = glob.glob(d + "/*/*counter_collection.csv")[0]
    acc = defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != name:
            continue
        k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        k = re.sub(r"^void ", "", k).split("(")[0]
        acc[k][0] += float(r["Counter_Value"])
        acc[k][1] += 1
    return acc


fe, wr = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(set(fe) | set(wr), key=lambda k: -(fe.get(k, [0, 0])[0] + wr.get(k, [0, 0])[0])):
    f, nf = fe.get(k, [0.0, 0])
    w, nw = wr.get(k, [0.0, 0])
    n = max(nf, nw, 1)
    out[k] = {"launches": n, "fetch_KiB_raw": round(f, 1), "write_KiB": round(w, 1),
              "fetch_bytes_per_launch_raw": round(f * 1024 / n), "fetch_bytes_per_launch_x2": round(2 * f * 1024 / n),
              "write_bytes_per_launch": round(w * 1024 / n)}
# how many passes over the workload the profiled command made (bench.py --steps 1 --warmup 1 = 2): the encoder's launch shape
# depends on the blocks in flight, so a consumer scales bytes per PASS to its own launches instead of trusting bytes per launch
out["_meta"] = {"passes": int(sys.argv[4]) if len(sys.argv) > 4 else 2}
json.dump(out, open(sys.argv[3], "w"), indent=1)
for k, v in [kv for kv in out.items() if kv[0] != "_meta"][:24]:
    print(f"{k[:34]:34s} n={v['launches']:4d} fetch/launch raw {v['fetch_bytes_per_launch_raw'] / 1e6:10.2f} MB  write/launch {v['write_bytes_per_launch'] / 1e6:10.2f} MB")
