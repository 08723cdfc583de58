#!/bin/bash
# A/B of the one-pass radix: forward BWT alone (rocprof kernel stats) and the bench loop
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/r04e; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_primitives.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -5) > $O/tests.log
cd /tmp
for v in 1 0; do
  rm -rf /tmp/kf$v
  JPK_ONESWEEP=$v rocprofv3 --kernel-trace -d /tmp/kf$v -o f -- python3 $R/tools/fwd_once.py text_survey 3 > /dev/null 2>&1
  python3 $R/tools/rocpd_stats.py /tmp/kf$v/f_results.db 3 2>&1 | head -14 > $O/fwd_stats_onesweep$v.txt
done
cd $R
for v in 1 0; do
  for rep in 1 2; do JPK_ONESWEEP=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-extras 2>/dev/null | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('onesweep=$v', j['value'], j['ms_per_step'])"; done
done > $O/bench_ab.log 2>&1
cat $O/tests.log $O/fwd_stats_onesweep1.txt $O/fwd_stats_onesweep0.txt $O/bench_ab.log
