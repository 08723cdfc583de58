#!/bin/bash
# Run on the MI355X box from the repo root:  bash tools/collect_profiles.sh <tag>
# Produces gpurun_out/<tag>/: PMC passes + summary, rocprofv3 kernel stats, bench lines.
set -u
TAG=${1:-r01_f}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
# 1. PMC passes (separate, counters only) of one compress+decompress pass
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$C
  rocprofv3 --pmc $C --output-format csv -d /tmp/pmc_$C -- python3 $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /tmp/pmc_$C.log 2>&1
  cp /tmp/pmc_$C/*/*counter_collection.csv "$OUT/$(echo $C | tr A-Z a-z)_counter_collection.csv" 2>/dev/null
done
python3 $REPO/tools/pmc_summary.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE $REPO/profiles/r01_pmc_traffic.json > "$OUT/pmc_summary.txt" 2>&1
cp $REPO/profiles/r01_pmc_traffic.json "$OUT/r01_pmc_traffic.json"
# 2. kernel trace + stats of the bench command
rm -rf /tmp/kt
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/bench_under_rocprof.json" 2>/tmp/kt.log
cp /tmp/kt/*/*kernel_stats.csv "$OUT/kernel_stats.csv" 2>/dev/null
# 3. the bench lines (traffic now comes from the fresh PMC summary)
cd $REPO
python3 bench.py --steps 5 --warmup 2 2>/dev/null | tail -1 > "$OUT/bench.json"
python3 bench.py --workload enwik9 --contexts 8 --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > "$OUT/bench_enwik9like_8ctx.json"
python3 bench.py --workload silesia --block-mib 256 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > "$OUT/bench_silesialike_256mib.json"
python3 tools/worst_cases.py > "$OUT/worst_cases.txt" 2>/dev/null
ls -la "$OUT"
