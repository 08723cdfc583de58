#!/bin/bash
set -u
TAG=${1:-r3q}
REPO=$PWD
OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export JAMPACK_CORPUS_CACHE=/tmp/jpk_corpus
export TMPDIR=/tmp
timeout 1500 python3 -m pytest tests -m gpu -x -q -s 2>&1 | grep -E "threads through|host copies|passed|failed|FAILED|Error" | tail -8 > $OUT/pytest.log
cd /tmp
rm -rf /tmp/ke
rocprofv3 --kernel-trace -d /tmp/ke -o e -- python3 $REPO/tools/enc_once.py text_survey > /dev/null 2>&1
python3 $REPO/tools/rocpd_stats.py /tmp/ke/e_results.db 3 > "$OUT/kernel_stats_ans_encode_64mib.txt" 2>&1
cd $REPO
python3 tools/stage_scaling.py 2>/dev/null | grep contexts > $OUT/stage_scaling.txt
python3 bench.py --steps 20 --warmup 5 --no-block-sizes 2>$OUT/bench.err | tail -1 > $OUT/bench.json
cat $OUT/pytest.log; head -16 $OUT/kernel_stats_ans_encode_64mib.txt; cat $OUT/stage_scaling.txt; python3 -c "
import json;d=json.load(open('$OUT/bench.json'));print(d['value'],d['ms_per_step'],d.get('stages_ms'))"
