#!/bin/bash
set -u
TAG=${1:-r3m}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export JAMPACK_CORPUS_CACHE=/tmp/jpk_corpus
timeout 1500 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $OUT/pytest.log
python3 tools/stage_scaling.py 2>/dev/null | grep contexts > $OUT/stage_scaling.txt
python3 bench.py --steps 20 --warmup 5 --no-block-sizes 2>$OUT/bench.err | tail -1 > $OUT/bench.json
cat $OUT/pytest.log $OUT/stage_scaling.txt; python3 -c "
import json;d=json.load(open('$OUT/bench.json'));print(d['value'],d['ms_per_step'],d.get('stages_ms'),d.get('decompress'))"
