#!/bin/bash
# round 3, GPU call 1: full GPU test suite + floor + radix scatter A/B + bench
set -u
REPO=$PWD
OUT=$REPO/gpurun_out/r3a
mkdir -p "$OUT"
export TMPDIR=/tmp
export JAMPACK_CORPUS_CACHE=/tmp/jpk_corpus
timeout 1500 python -m pytest tests -m gpu -x -q > "$OUT/pytest.log" 2>&1
echo "pytest rc=$?" >> "$OUT/pytest.log"
timeout 120 tools/_bin/sa_floor > "$OUT/sa_floor.txt" 2>&1
for V in 0 1; do
  JPK_RS_STAGED=$V timeout 300 python3 tools/stage_scaling.py 2>/dev/null | grep contexts > "$OUT/stage_scaling_staged$V.txt"
done
cd /tmp
for V in 0 1; do
  rm -rf /tmp/kf$V
  JPK_RS_STAGED=$V timeout 300 rocprofv3 --kernel-trace -d /tmp/kf$V -o f -- python3 $REPO/tools/fwd_once.py text_survey 3 > /dev/null 2>&1
  python3 $REPO/tools/rocpd_stats.py /tmp/kf$V/f_results.db 3 > "$OUT/kernel_stats_forward_bwt_64mib_staged$V.txt" 2>&1
done
rm -rf /tmp/ke
timeout 300 rocprofv3 --kernel-trace -d /tmp/ke -o e -- python3 $REPO/tools/enc_once.py text_survey > /dev/null 2>&1
python3 $REPO/tools/rocpd_stats.py /tmp/ke/e_results.db 3 > "$OUT/kernel_stats_ans_encode_64mib.txt" 2>&1
cd $REPO
timeout 600 python3 bench.py --steps 20 --warmup 5 2>"$OUT/bench.err" | tail -1 > "$OUT/bench.json"
JPK_RS_STAGED=0 timeout 300 python3 bench.py --steps 20 --warmup 5 --no-extras 2>/dev/null | tail -1 > "$OUT/bench_staged0.json"
ls -la "$OUT"
tail -5 "$OUT/pytest.log"
