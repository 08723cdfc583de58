#!/bin/bash
set -u
REPO=$PWD
OUT=$REPO/gpurun_out/r3h
mkdir -p "$OUT"
export TMPDIR=/tmp
export JAMPACK_CORPUS_CACHE=/tmp/jpk_corpus
timeout 900 python -m pytest tests/test_gpu_primitives.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py -q -x 2>&1 | tail -3 > "$OUT/pytest_parity.log"
for V in 0 24576 40960 65536; do
  JPK_OCC_MEM=$V timeout 300 python3 bench.py --steps 20 --warmup 5 --no-extras 2>/dev/null | tail -1 > "$OUT/bench_occ$V.json"
done
cd /tmp
rm -rf /tmp/kf
timeout 300 rocprofv3 --kernel-trace -d /tmp/kf -o f -- python3 $REPO/tools/fwd_once.py text_survey 3 > /dev/null 2>&1
python3 $REPO/tools/rocpd_stats.py /tmp/kf/f_results.db 3 > "$OUT/kernel_stats_forward_bwt_64mib.txt" 2>&1
# kernel trace of the bench loop itself (4 blocks in flight): which kernels stretch
rm -rf /tmp/kt
timeout 600 rocprofv3 --kernel-trace -d /tmp/kt -o kt -- python3 $REPO/bench.py --steps 10 --warmup 3 --no-extras > /dev/null 2>&1
python3 $REPO/tools/rocpd_stats.py /tmp/kt/kt_results.db > "$OUT/kernel_stats_bench_loop.txt" 2>&1
python3 $REPO/tools/rocpd_busy.py /tmp/kt/kt_results.db > "$OUT/busy_bench_loop.txt" 2>&1
cd $REPO
cat "$OUT/pytest_parity.log"
for V in 0 24576 40960 65536; do python3 -c "
import json
try:
    d=json.load(open('$OUT/bench_occ$V.json')); print('occ $V', d['value'], d['ms_per_step'])
except Exception as e: print('occ $V ERR', e)"; done
