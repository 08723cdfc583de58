#!/bin/bash
# kernel stats of three rANS encodes of a 64 MiB text image (per repetition) + the default bench line twice:  bash tools/enc_stats.sh <outdir>
export TMPDIR=/tmp; R=$PWD; O=$R/${1:-gpurun_out/enc_stats}; mkdir -p $O
cd /tmp; rm -rf /tmp/ke
rocprofv3 --kernel-trace -d /tmp/ke -o e -- python3 $R/tools/enc_once.py text_survey > /dev/null 2>&1
python3 $R/tools/rocpd_stats.py /tmp/ke/e_results.db 3 2>&1 | grep -v "k_seg\|k_gather\|k_rs_\|k_r0\|k_lg\|k_cmp\|k_win\|k_tab\|k_scan\|k_bwt\|k_run" | head -30 > $O/enc_stats.txt
cd $R
for r in 1 2 3; do timeout 200 python bench.py --steps 20 --warmup 5 --no-extras 2>/dev/null | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'])"; done > $O/bench.txt
cat $O/enc_stats.txt $O/bench.txt
