#!/bin/bash
# enwik9-like compress throughput against blocks in flight and hardware queues (corpus generated once)
cd "$(dirname "$0")/.."
export JAMPACK_CORPUS_CACHE=/tmp/jpk_corpus
QS=${QS:-"16 32 64"}; CS=${CS:-"4 8 12 15"}
for q in $QS; do for c in $CS; do
  GPU_MAX_HW_QUEUES=$q python bench.py --workload enwik9 --contexts $c --steps 2 --warmup 1 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('queues $q contexts $c:', d['value'], 'MB/s', d['ms_per_step'], 'ms')"
done; done
