#!/bin/bash
bash tools/collect_profiles.sh r03b > /dev/null 2>&1
export JAMPACK_CORPUS_CACHE=/tmp/jpk_corpus
for C in 3 4 5 6; do python3 bench.py --contexts $C --steps 16 --warmup 4 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('contexts', $C, d['value'], d['ms_per_step'])"; done > gpurun_out/r03b/contexts_sweep.txt
cat gpurun_out/r03b/contexts_sweep.txt
python3 -c "
import json;d=json.load(open('gpurun_out/r03b/bench.json'));print(d['value'],d['ms_per_step'],d.get('stages_ms'))"
