#!/bin/bash
# Run on the MI355X box from the repo root:  bash tools/collect_profiles.sh <tag>
# Produces gpurun_out/<tag>/: PMC passes + summary, rocprofv3 kernel stats of the bench command, bench lines, worst cases.
# (counters and traces in separate runs; the program itself after `--`)
set -u
TAG=${1:-r06}
REPO=$PWD
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
export JAMPACK_CORPUS_CACHE=/tmp/jpk_corpus
cd /tmp
# 0. the bench line first, on the untouched box: after the counter-collection passes below the host-buffer extras of a bench run on the
#    same box were 10-20x slow on two of three boxes (SDMA copies; GPU-resident figures unaffected) -- profiles/r04_bench_after_rocprof_passes.json
( cd $REPO && python3 bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > "$OUT/bench.json" )
# 1. PMC passes (separate, counters only) of one compress step of the default workload
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$C
  rocprofv3 --pmc $C --output-format csv -d /tmp/pmc_$C -- python3 $REPO/bench.py --steps 1 --warmup 1 --no-extras > /tmp/pmc_$C.log 2>&1
  cp /tmp/pmc_$C/*/*counter_collection.csv "$OUT/$(echo $C | tr A-Z a-z)_counter_collection.csv" 2>/dev/null
done
python3 $REPO/tools/pmc_summary.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE "$OUT/pmc_traffic.json" > "$OUT/pmc_summary.txt" 2>&1
# 2. kernel trace + stats of the bench command
rm -rf /tmp/kt
rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python3 $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline > "$OUT/bench_under_rocprof.json" 2>/tmp/kt.log
python3 $REPO/tools/rocpd_stats.py /tmp/kt/kt_results.db > "$OUT/kernel_stats.txt" 2>&1
# 3. stage-level traces of one 64 MiB block (per-rep numbers)
rm -rf /tmp/kf /tmp/ke
rocprofv3 --kernel-trace -d /tmp/kf -o f -- python3 $REPO/tools/fwd_once.py text_survey 3 > /dev/null 2>&1
python3 $REPO/tools/rocpd_stats.py /tmp/kf/f_results.db 4 > "$OUT/kernel_stats_forward_bwt_64mib.txt" 2>&1
rocprofv3 --kernel-trace -d /tmp/ke -o e -- python3 $REPO/tools/enc_once.py text_survey > /dev/null 2>&1
python3 $REPO/tools/rocpd_stats.py /tmp/ke/e_results.db 3 > "$OUT/kernel_stats_ans_encode_64mib.txt" 2>&1
python3 $REPO/tools/rocpd_timeline.py /tmp/ke/e_results.db k_density > "$OUT/timeline_ans_encode_64mib.txt" 2>&1
# 4. the bench lines
cd $REPO
cp "$OUT/pmc_traffic.json" profiles/${TAG}_pmc_traffic.json 2>/dev/null     # bench.py reads the traffic of the dominant kernel from here (step 0 used the file of the collection before)
python3 bench.py --workload enwik9 --steps 2 --warmup 1 --no-extras 2>/dev/null | tail -1 > "$OUT/bench_enwik9like.json"
python3 bench.py --workload silesia --block-mib 256 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > "$OUT/bench_silesialike_256mib.json"
python3 tools/worst_cases.py > "$OUT/worst_cases.txt" 2>/dev/null
NLIST=1,4,8,16,24,32,48,64 python3 tools/dec_scaling.py batch > "$OUT/decode_batch_scaling.txt" 2>/dev/null
python3 tools/stage_scaling.py 2>/dev/null | grep contexts > "$OUT/stage_scaling.txt"
python3 tools/chunk_lens.py 2>/dev/null | grep -E "chunks|k_dec_" > "$OUT/decode_chain_lengths.txt"
ls -la "$OUT"
python3 tools/block_sizes.py 1,8,64,128 2>/dev/null > "$OUT/block_sizes.txt"
python3 tools/batch_compress.py 2>/dev/null | grep "in flight" > "$OUT/blocks_compress_call.txt"
# 5. round 3: the floor of the suffix sort, the drop-in block loop with 16 threads, the N > 1 code path on one GPU, SQ counters
( echo "# headline block (text_survey, order-2 context codes: round counts of tools/fwd_once.py)"; tools/_bin/sa_floor 67108800 27281026 48909
  echo "# the same block with the order-0 variable-length keys (JPK_KEY_ORDER=0)"; tools/_bin/sa_floor 67108800 41245987 5028889 321
  echo "# the same block with round 4's fixed-width keys (11 bytes)"; tools/_bin/sa_floor 67108800 47350511 8839248 3530
  echo "# wide-alphabet block (text_wide, order-2 context codes)"; tools/_bin/sa_floor 67108800 20205395 45204
  echo "# the same block with the order-0 variable-length keys"; tools/_bin/sa_floor 67108800 38943552 4138392 3586
  echo "# the same block with fixed-width keys (7 bytes)"; tools/_bin/sa_floor 67108800 56766712 23323321 420103 ) > "$OUT/sa_floor.txt" 2>&1
python3 - <<'PY'
import sys
sys.path.insert(0, ".")
from jampack_amd import corpus
d, _ = corpus.load_or_make("enwik9", start=0, count=4 * (64 << 20))
with open("/tmp/pipe_in16.bin", "wb") as f:
    for _ in range(4):
        d.tofile(f)
PY
make -C jampack_amd/csrc/shim > /dev/null
( jampack_amd/csrc/shim/jam_block_pipeline /tmp/pipe_in16.bin 64 8 | tail -1; jampack_amd/csrc/shim/jam_block_pipeline /tmp/pipe_in16.bin 64 16 | tail -1; JPK_COMBINE_US=-1 jampack_amd/csrc/shim/jam_block_pipeline /tmp/pipe_in16.bin 64 16 | tail -1 | sed 's/$/   [JPK_COMBINE_US=-1: combiner off]/' ) > "$OUT/block_pipeline_threads.txt" 2>&1
JPK_BENCH_ONE_GPU=1 python3 bench.py --gpus 2 --steps 4 --warmup 2 --no-extras --contexts 2 2>/dev/null | tail -1 > "$OUT/bench_two_ranks_one_gpu.json"
JPK_BENCH_ONE_GPU=1 python3 bench.py --gpus 2 --steps 2 --warmup 1 --no-extras --contexts 2 --workload enwik9 --limit-bytes 402653184 2>/dev/null | tail -1 > "$OUT/bench_two_ranks_one_gpu_enwik9_384mib.json"
bash tools/pmc_sq.sh gpurun_out/$TAG/sq_fwd fwd > /dev/null 2>&1
bash tools/pmc_sq.sh gpurun_out/$TAG/sq_dec dec > /dev/null 2>&1
python3 tools/dec_once.py text_survey 2 2>/dev/null | tail -1 > "$OUT/decode_once.txt"
tools/_bin/issuetest > "$OUT/issuetest.txt" 2>&1
# 6. round 4: small blocks through one call (a fresh process)
python3 tools/small_blocks.py 1,8,64 4,8,16 2>/dev/null | grep blocks > "$OUT/small_blocks.txt"
# 7. round 5: variable-length keys against the fixed-width ones, the pair rule against plain doubling (same box, alternating)
if [ "${JPK_COLLECT_AB:-0}" = 1 ]; then      # (round 5's A/B sections: unchanged code in round 6, ~10 GPU-minutes; JPK_COLLECT_AB=1 repeats them)
( echo "# bench.py --steps 12 --warmup 3 --no-extras, same box, alternating: variable-length keys (default) against JPK_VARKEYS=0 (round 4's alphabet-packed fixed-width keys)"
  for i in 1 2 3; do for v in 1 0; do
    echo -n "headline (28 byte values) JPK_VARKEYS=$v  "; JPK_VARKEYS=$v python3 bench.py --steps 12 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | grep -o '"value": [0-9.]*' | head -1
    echo -n "wide (207 byte values)    JPK_VARKEYS=$v  "; JPK_VARKEYS=$v python3 bench.py --workload enwik8-wide --steps 12 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | grep -o '"value": [0-9.]*' | head -1
  done; done
  echo "# forward BWT of ONE 64 MiB block at a time (tools/fwd_once.py), wall clock per block"
  for k in text_survey text_wide; do for v in 1 0; do echo -n "$k JPK_VARKEYS=$v  "; JPK_VARKEYS=$v python3 tools/fwd_once.py $k 3 2>/dev/null | tail -1; done; done
  echo "# ... and by the order of the code (JPK_KEY_ORDER: 0 = the round's first form, 1, 2 = default)"
  for k in text_survey text_wide; do for v in 0 1 2; do echo -n "$k JPK_KEY_ORDER=$v  "; JPK_KEY_ORDER=$v python3 tools/fwd_once.py $k 10 2>/dev/null | tail -1; done; done ) > "$OUT/var_keys.txt"
( echo "# the pair rule (k_pair_*) against plain prefix doubling (JPK_PAIR_SHIFT=-1) and without the repair walk (JPK_PAIR_REPAIR=0): forward BWT of one 64 MiB block"
  for k in repeat silesia runs text; do
    echo -n "$k default            "; python3 tools/fwd_once.py $k 3 2>/dev/null | tail -1
    echo -n "$k JPK_PAIR_REPAIR=0  "; JPK_PAIR_REPAIR=0 python3 tools/fwd_once.py $k 3 2>/dev/null | tail -1
    echo -n "$k JPK_PAIR_SHIFT=-1  "; JPK_PAIR_SHIFT=-1 python3 tools/fwd_once.py $k 3 2>/dev/null | tail -1
  done
  echo "# config 5: bench.py --workload silesia --block-mib 256 --steps 4 --warmup 1 --no-extras"
  for e in "" "JPK_PAIR_SHIFT=-1" "JPK_VARKEYS=0" "JPK_PAIR_SHIFT=-1 JPK_VARKEYS=0"; do echo -n "silesia-like 212 MB [$e]  "; env $e python3 bench.py --workload silesia --block-mib 256 --steps 4 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | grep -o '"value": [0-9.]*' | head -1; done ) > "$OUT/pair_rule.txt"
fi
rm -rf /tmp/kfw
( cd /tmp; rocprofv3 --kernel-trace -d /tmp/kfw -o f -- python3 $REPO/tools/fwd_once.py text_wide 3 > /dev/null 2>&1; python3 $REPO/tools/rocpd_stats.py /tmp/kfw/f_results.db 4 > "$OUT/kernel_stats_forward_bwt_64mib_wide.txt" 2>&1 )
ls -la "$OUT"
# 8. round 6: structured inputs (cliffs), blocks above 2^28 bytes, contexts in flight on the bench line, the ticket micro-benchmark
python3 tools/structured_inputs.py > "$OUT/structured_inputs.txt" 2>/dev/null
( for c in 8 10 8 10 8 10; do echo -n "contexts $c "; python3 bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --contexts $c 2>/dev/null | grep -o '"value": [0-9.]*' | head -1; done ) > "$OUT/blocks_in_flight.txt"
( echo "# one 300 MiB block (above 2^28 bytes: the slots' depths in their own array), variable-length keys (default) against JPK_VARKEYS=0 (round 4's fixed-width keys, what such blocks got before round 6)"
  python3 tools/big_block.py $((300 << 20)) 2>/dev/null | tail -3
  echo "# JPK_VARKEYS=0"; JPK_VARKEYS=0 python3 tools/big_block.py $((300 << 20)) 2>/dev/null | tail -3 ) > "$OUT/big_block_300mib.txt"
tools/_bin/tickettest > "$OUT/tickettest.txt" 2>&1
ls -la "$OUT"
