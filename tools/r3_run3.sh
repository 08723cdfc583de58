#!/bin/bash
# round 3, GPU call 3: BWT byte carried with the suffix + 10-instruction rANS step: tests, stage profiles, contexts sweep
set -u
REPO=$PWD
OUT=$REPO/gpurun_out/r3g
mkdir -p "$OUT"
export TMPDIR=/tmp
export JAMPACK_CORPUS_CACHE=/tmp/jpk_corpus
timeout 1800 python -m pytest tests -m gpu -x -q > "$OUT/pytest.log" 2>&1
echo "pytest rc=$?" >> "$OUT/pytest.log"
timeout 300 python3 tools/stage_scaling.py 2>/dev/null | grep contexts > "$OUT/stage_scaling.txt"
cd /tmp
rm -rf /tmp/kf /tmp/ke
timeout 300 rocprofv3 --kernel-trace -d /tmp/kf -o f -- python3 $REPO/tools/fwd_once.py text_survey 3 > /dev/null 2>&1
python3 $REPO/tools/rocpd_stats.py /tmp/kf/f_results.db 3 > "$OUT/kernel_stats_forward_bwt_64mib.txt" 2>&1
timeout 300 rocprofv3 --kernel-trace -d /tmp/ke -o e -- python3 $REPO/tools/enc_once.py text_survey > /dev/null 2>&1
python3 $REPO/tools/rocpd_stats.py /tmp/ke/e_results.db 3 > "$OUT/kernel_stats_ans_encode_64mib.txt" 2>&1
cd $REPO
for C in 3 4 5 6 8; do
  timeout 300 python3 bench.py --steps 20 --warmup 5 --no-extras --contexts $C 2>/dev/null | tail -1 > "$OUT/bench_ctx$C.json"
done
timeout 600 python3 bench.py --steps 20 --warmup 5 --no-block-sizes 2>"$OUT/bench.err" | tail -1 > "$OUT/bench.json"
timeout 300 python3 tools/worst_cases.py > "$OUT/worst_cases.txt" 2>/dev/null
ls -la "$OUT"
tail -5 "$OUT/pytest.log"
