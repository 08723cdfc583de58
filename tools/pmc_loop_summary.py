#!/usr/bin/env python3
"""Per-kernel sums of the counter passes of tools/pmc_loop.sh + derived ratios.
   python tools/pmc_loop_summary.py <tmpdir holding pl_*/> <outdir>"""
import csv
import glob
import json
import os
import re
import sys

tmp, out = sys.argv[1], sys.argv[2]
tot, passes = {}, []
for d in sorted(glob.glob(tmp + "/pl_*")):
    if not os.path.isdir(d) or d.endswith("pl_kt"):
        continue
    files = glob.glob(d + "/*/*counter_collection.csv")
    if not files:
        print(f"# pass {os.path.basename(d)}: no counter file (counters refused?)")
        continue
    passes.append(os.path.basename(d))
    for f in files:
        for r in csv.DictReader(open(f)):
            k = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"]).split("(")[0]
            k = re.sub(r"<.*", "", k)
            t = tot.setdefault(k, {})
            t[r["Counter_Name"]] = t.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            t["_n"] = max(t.get("_n", 0), 0)
for nm in ("sq1", "tcc1", "free", "trace"):
    for f in glob.glob(out + f"/bench_under_pmc_{nm}.json") + glob.glob(out + f"/bench_{nm}.json") + glob.glob(out + f"/bench_under_{nm}.json"):
        try:
            j = json.loads(open(f).read().strip().splitlines()[-1])
            print(f"# {os.path.basename(f)}: value {j['value']} MB/s, {j['ms_per_step']} ms per step")
        except Exception as e:  # noqa: BLE001
            print(f"# {os.path.basename(f)}: unreadable ({e})")
print("# passes:", " ".join(passes))
print("# sums over all launches of `bench.py --steps 10 --warmup 3 --no-extras` (13 passes over 100 MB, 4 blocks in flight when free running)")


def g(t, n):
    return t.get(n, 0.0)


hdr = ["kernel", "wave_cyc", "wait_any%", "wait_inst%", "act_any%", "valu%", "lds%", "ldsconf%", "VALU_inst", "LDS_inst", "VMEM_rd", "VMEM_wr",
       "EA_rd_MB", "EA_wr_MB", "L2hit%", "tcp_stall"]
print(" ".join(f"{h:>11s}" if i else f"{h:22s}" for i, h in enumerate(hdr)))
for k, t in sorted(tot.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:36]:
    wc = g(t, "SQ_WAVE_CYCLES") or 1.0
    idx = g(t, "SQ_LDS_IDX_ACTIVE") or 1.0
    hm = (g(t, "TCC_HIT_sum") + g(t, "TCC_MISS_sum")) or 1.0
    row = [f"{k[:22]:22s}", f"{wc:11.3g}", f"{100 * g(t, 'SQ_WAIT_ANY') / wc:11.1f}", f"{100 * g(t, 'SQ_WAIT_INST_ANY') / wc:11.1f}",
           f"{100 * g(t, 'SQ_ACTIVE_INST_ANY') / wc:11.1f}", f"{100 * g(t, 'SQ_ACTIVE_INST_VALU') / wc:11.1f}", f"{100 * g(t, 'SQ_ACTIVE_INST_LDS') / wc:11.1f}",
           f"{100 * g(t, 'SQ_LDS_BANK_CONFLICT') / idx:11.1f}", f"{g(t, 'SQ_INSTS_VALU'):11.3g}", f"{g(t, 'SQ_INSTS_LDS'):11.3g}",
           f"{g(t, 'SQ_INSTS_VMEM_RD'):11.3g}", f"{g(t, 'SQ_INSTS_VMEM_WR'):11.3g}", f"{g(t, 'TCC_EA0_RDREQ_sum') * 64 / 1e6:11.1f}",
           f"{g(t, 'TCC_EA0_WRREQ_sum') * 64 / 1e6:11.1f}", f"{100 * g(t, 'TCC_HIT_sum') / hm:11.1f}", f"{g(t, 'TCP_PENDING_STALL_CYCLES_sum'):11.3g}"]
    print(" ".join(row))
print()
print("# raw sums per kernel (every counter collected)")
for k, t in sorted(tot.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:36]:
    print(k, " ".join(f"{n}={v:.4g}" for n, v in sorted(t.items()) if not n.startswith("_")))
