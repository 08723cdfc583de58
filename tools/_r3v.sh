#!/bin/bash
export JAMPACK_CORPUS_CACHE=/tmp/jpk_corpus
mkdir -p gpurun_out/r3v
python3 bench.py --steps 20 --warmup 5 --no-block-sizes 2>/dev/null | tail -1 > gpurun_out/r3v/bench.json
NLIST=4,8,16,64 python3 tools/dec_scaling.py batch 2>/dev/null | tee gpurun_out/r3v/dec.txt
python3 -c "
import json; d=json.load(open('gpurun_out/r3v/bench.json')); x=d['decompress']; print(d['value'], x['value'], x['MBps_by_passes_in_flight'], x['one_pass_at_a_time_MBps'])"
