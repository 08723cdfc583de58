#!/bin/bash
export JAMPACK_CORPUS_CACHE=/tmp/jpk_corpus
mkdir -p gpurun_out/r3w
run() { python3 bench.py --steps 8 --warmup 3 --no-block-sizes --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); x=d['decompress']; print('$1', d['value'], x['value'], x['MBps_by_passes_in_flight'], x['one_pass_at_a_time_MBps'])"; }
cp jampack_amd/libjampack_amd.so /tmp/cur.so
run cur; run cur
cp tools/_bin/libjampack_prev_decrank.so jampack_amd/libjampack_amd.so
run prev; run prev
cp /tmp/cur.so jampack_amd/libjampack_amd.so
run cur
