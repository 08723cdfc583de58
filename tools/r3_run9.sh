#!/bin/bash
set -u
REPO=$PWD
OUT=$REPO/gpurun_out/r3i
mkdir -p "$OUT"
export TMPDIR=/tmp
export JAMPACK_CORPUS_CACHE=/tmp/jpk_corpus
timeout 1800 python -m pytest tests -m gpu -x -q > "$OUT/pytest.log" 2>&1
echo "pytest rc=$?" >> "$OUT/pytest.log"
for cfg in "JPK_RS_STAGED=1 JPK_SEG_OCC=4" "JPK_RS_STAGED=0 JPK_SEG_OCC=4" "JPK_RS_STAGED=1 JPK_SEG_OCC=5" "JPK_RS_STAGED=0 JPK_SEG_OCC=5"; do
  tag=$(echo $cfg | tr ' =' '__')
  for rep in 1 2; do
    env $cfg timeout 300 python3 bench.py --steps 20 --warmup 5 --no-extras 2>/dev/null | tail -1 > "$OUT/bench_${tag}_$rep.json"
  done
  env $cfg timeout 300 python3 tools/stage_scaling.py 2>/dev/null | grep -E "fwd: (1|3)" > "$OUT/fwd_${tag}.txt"
done
tail -3 "$OUT/pytest.log"
for f in $OUT/bench_*.json; do python3 -c "
import json
try:
    d=json.load(open('$f')); print('$f'.split('/')[-1], d['value'], d['ms_per_step'])
except Exception as e: print('$f ERR', e)"; done
cat $OUT/fwd_*.txt
