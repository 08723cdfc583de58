#!/usr/bin/env python3
"""print the kernel timeline of the last entropy-encode call from a rocprofv3 --kernel-trace csv"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
marker = sys.argv[2] if len(sys.argv) > 2 else "k_density"
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
start = idx[-1]
t0 = int(rows[start]["Start_Timestamp"])
for r in rows[start:]:
    import re
    m = re.search(r"(k_\w+)", r["Kernel_Name"])
    n = (m.group(1) if m else r["Kernel_Name"])[:24]
    s = (int(r["Start_Timestamp"]) - t0) / 1e6
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    print("%-24s q=%3s start=%8.3f ms dur=%8.3f grid=%sx%sx%s" % (n, r["Queue_Id"], s, d, r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"]))
