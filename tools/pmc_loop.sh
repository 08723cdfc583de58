#!/bin/bash
# Counters per kernel IN THE TIMED LOOP'S OWN SHAPE (bench.py --steps 10 --warmup 3 --no-extras, 4 blocks in flight):
#   bash tools/pmc_loop.sh <outdir>     -> <outdir>/loop_counters.txt
# Counter passes only (no trace domains next to --pmc); the program itself after `--`.  A kernel-trace of the same
# command WITHOUT counters rides along so that the per-kernel durations under counter collection can be compared with
# the free-running loop (counter collection serialises dispatches on this stack: the file says so when it sees it).
set -u
REPO=$PWD
OUT=$REPO/${1:-gpurun_out/pmc_loop}
mkdir -p "$OUT"
export TMPDIR=/tmp
export JAMPACK_CORPUS_CACHE=/tmp/jpk_corpus
CMD="python3 $REPO/bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline"
cd /tmp
rocprofv3 -L 2>/dev/null | grep -o -E "\b(SQ|TCC|TCP|TA|TD|GRBM)_[A-Z0-9_]+\b" | sort -u > "$OUT/available_counters.txt"
run() { # name counters...
  local n=$1; shift
  rm -rf /tmp/pl_$n
  rocprofv3 --pmc "$@" --output-format csv -d /tmp/pl_$n -- $CMD > "$OUT/bench_under_pmc_$n.json" 2> /tmp/pl_$n.log
  tail -3 /tmp/pl_$n.log > "$OUT/pmc_$n.stderr_tail.txt"
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
run sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES
run sq3 SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_INSTS_SMEM SQ_ACTIVE_INST_FLAT
run tcc1 TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum
run tcc2 TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum TCC_ATOMIC_sum
run tcp1 TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
rm -rf /tmp/pl_kt
rocprofv3 --kernel-trace -d /tmp/pl_kt -o kt -- $CMD > "$OUT/bench_under_trace.json" 2> /tmp/pl_kt.log
python3 $REPO/tools/rocpd_stats.py /tmp/pl_kt/kt_results.db > "$OUT/kernel_stats_loop.txt" 2>&1
$CMD > "$OUT/bench_free.json" 2>/dev/null
python3 $REPO/tools/pmc_loop_summary.py /tmp "$OUT" > "$OUT/loop_counters.txt" 2>&1
ls -la "$OUT"
