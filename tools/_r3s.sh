#!/bin/bash
set -u
TAG=${1:-r3s}
REPO=$PWD
OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export JAMPACK_CORPUS_CACHE=/tmp/jpk_corpus
export TMPDIR=/tmp
timeout 1500 python3 -m pytest tests -m gpu -x -q -s 2>&1 | grep -E "threads through|host copies|passed|failed|FAILED|Error" | tail -8 > $OUT/pytest.log
NLIST=1,4,8,16,24,32,48,64 python3 tools/dec_scaling.py batch > $OUT/decode_batch_scaling.txt 2>/dev/null
cd /tmp
rm -rf /tmp/kd64
NLIST=64 rocprofv3 --kernel-trace -d /tmp/kd64 -o d -- python3 $REPO/tools/dec_scaling.py batch > $OUT/dec64.log 2>&1
python3 $REPO/tools/rocpd_stats.py /tmp/kd64/d_results.db 2 > $OUT/kernel_stats_decode_batch64.txt 2>&1
cd $REPO
cat $OUT/pytest.log $OUT/decode_batch_scaling.txt; head -12 $OUT/kernel_stats_decode_batch64.txt
