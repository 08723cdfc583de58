#!/bin/bash
set -u
REPO=$PWD
OUT=$REPO/gpurun_out/r3k
mkdir -p "$OUT"
export TMPDIR=/tmp
export JAMPACK_CORPUS_CACHE=/tmp/jpk_corpus
timeout 900 python -m pytest tests/test_gpu_stages.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_random_diff.py tests/test_gpu_batch.py -q -x 2>&1 | tail -3 > "$OUT/pytest_parity.log"
for rep in 1 2; do
  timeout 300 python3 bench.py --steps 20 --warmup 5 --no-extras 2>/dev/null | tail -1 > "$OUT/bench_$rep.json"
done
timeout 300 python3 tools/stage_scaling.py 2>/dev/null | grep -E "fwd" > "$OUT/fwd.txt"
cd /tmp; rm -rf /tmp/kf
timeout 300 rocprofv3 --kernel-trace -d /tmp/kf -o f -- python3 $REPO/tools/fwd_once.py text_survey 3 > /dev/null 2>&1
python3 $REPO/tools/rocpd_stats.py /tmp/kf/f_results.db 3 > "$OUT/kernel_stats_forward_bwt_64mib.txt" 2>&1
rm -rf /tmp/ke
timeout 300 rocprofv3 --kernel-trace -d /tmp/ke -o e -- python3 $REPO/tools/enc_once.py text_survey > /dev/null 2>&1
python3 $REPO/tools/rocpd_stats.py /tmp/ke/e_results.db 3 > "$OUT/kernel_stats_ans_encode_64mib.txt" 2>&1
cd $REPO
timeout 300 python3 tools/stage_scaling.py 2>/dev/null | grep -E "enc|both" > "$OUT/enc.txt"
cat "$OUT/pytest_parity.log"
head -12 "$OUT/kernel_stats_ans_encode_64mib.txt"; cat "$OUT/enc.txt"
for f in $OUT/bench_*.json; do python3 -c "
import json
try:
    d=json.load(open('$f')); print('$f'.split('/')[-1], d['value'], d['ms_per_step'])
except Exception as e: print('$f ERR', e)"; done
cat $OUT/fwd.txt; head -8 $OUT/kernel_stats_forward_bwt_64mib.txt
