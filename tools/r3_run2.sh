#!/bin/bash
# round 3, GPU call 2: GPU test suite (combiner, enqueue-only inverse BWT), decode batch scaling, bench
set -u
REPO=$PWD
OUT=$REPO/gpurun_out/r3b
mkdir -p "$OUT"
export TMPDIR=/tmp
export JAMPACK_CORPUS_CACHE=/tmp/jpk_corpus
timeout 1800 python -m pytest tests -m gpu -x -q > "$OUT/pytest.log" 2>&1
echo "pytest rc=$?" >> "$OUT/pytest.log"
NLIST=1,4,16,32,64 timeout 600 python3 tools/dec_scaling.py batch > "$OUT/decode_batch_scaling.txt" 2>/dev/null
timeout 300 python3 tools/stage_scaling.py 2>/dev/null | grep contexts > "$OUT/stage_scaling.txt"
timeout 600 python3 bench.py --steps 20 --warmup 5 --no-block-sizes 2>"$OUT/bench.err" | tail -1 > "$OUT/bench.json"
ls -la "$OUT"
tail -5 "$OUT/pytest.log"
cat "$OUT/decode_batch_scaling.txt"
