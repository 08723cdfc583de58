#!/bin/bash
# SQ counters per kernel of one forward BWT (or, with `enc`, three rANS encodes) of a 64 MiB text block (counters only,
# two passes of <= 8):   bash tools/pmc_sq.sh <outdir> [fwd|enc]        -> <outdir>/sq_summary.txt
set -u
REPO=$PWD
OUT=$REPO/${1:-gpurun_out/pmc_sq}
WHAT=${2:-fwd}
if [ "$WHAT" = enc ]; then PROG="$REPO/tools/enc_once.py text_survey"; elif [ "$WHAT" = dec ]; then PROG="$REPO/tools/dec_once.py text_survey 1"; else PROG="$REPO/tools/fwd_once.py text_survey 1"; fi
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/sq1 /tmp/sq2
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d /tmp/sq1 -- python3 $PROG > /tmp/sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES --output-format csv -d /tmp/sq2 -- python3 $PROG > /tmp/sq2.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, re, sys
out = sys.argv[1]
tot = {}
for d in ("/tmp/sq1", "/tmp/sq2"):
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"]).split("(")[0]
            t = tot.setdefault(k, {})
            t[r["Counter_Name"]] = t.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            t["_n_" + r["Counter_Name"]] = t.get("_n_" + r["Counter_Name"], 0) + 1
names = ["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM",
         "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAVES"]
with open(out + "/sq_summary.txt", "w") as fo:
    fo.write("kernel " + " ".join(names) + "   (sums over all launches of the run)\n")
    for k, t in sorted(tot.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:28]:
        fo.write(f"{k:24s} " + " ".join(f"{t.get(n, 0):.3g}" for n in names) + "\n")
print(open(out + "/sq_summary.txt").read())
PY
