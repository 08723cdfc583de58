export TMPDIR=/tmp; R=$PWD; cd /tmp
rm -rf /tmp/kf; rocprofv3 --kernel-trace -d /tmp/kf -o f -- python3 $R/tools/fwd_once.py text_survey 3 > /dev/null 2>&1
python3 $R/tools/rocpd_stats.py /tmp/kf/f_results.db 3 > $R/gpurun_out/ks_fwd_packed.txt 2>&1
rm -rf /tmp/kl; rocprofv3 --kernel-trace -d /tmp/kl -o l -- python3 $R/bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline > /dev/null 2>&1
python3 $R/tools/rocpd_stats.py /tmp/kl/l_results.db > $R/gpurun_out/ks_loop_packed.txt 2>&1
cd $R
python3 tools/worst_cases.py 2>/dev/null > gpurun_out/worst_packed.txt
python3 tools/small_blocks.py 1,8 8,16 2>/dev/null | grep blocks > gpurun_out/small_packed.txt
python3 bench.py --workload silesia --block-mib 256 --steps 2 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > gpurun_out/silesia_packed.json
python3 bench.py --workload enwik8-phrase --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > gpurun_out/phrase_packed.json
