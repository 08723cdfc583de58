#!/bin/bash
set -u
REPO=$PWD
OUT=$REPO/gpurun_out/r3r; mkdir -p $OUT
export JAMPACK_CORPUS_CACHE=/tmp/jpk_corpus
export TMPDIR=/tmp
cd /tmp
for N in 64 16; do
rm -rf /tmp/kd$N
NLIST=$N rocprofv3 --kernel-trace -d /tmp/kd$N -o d -- python3 $REPO/tools/dec_scaling.py batch > $OUT/dec$N.log 2>&1
python3 $REPO/tools/rocpd_stats.py /tmp/kd$N/d_results.db 2 > $OUT/kernel_stats_dec$N.txt 2>&1
python3 $REPO/tools/rocpd_timeline.py /tmp/kd$N/d_results.db k_dec_headers > $OUT/timeline_dec$N.txt 2>&1
done
tail -2 $OUT/dec64.log; head -30 $OUT/kernel_stats_dec64.txt; tail -60 $OUT/timeline_dec64.txt
