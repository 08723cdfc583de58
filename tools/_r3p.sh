#!/bin/bash
set -u
OUT=gpurun_out/r3p; mkdir -p $OUT
export JAMPACK_CORPUS_CACHE=/tmp/jpk_corpus
{ nproc; grep -c processor /proc/cpuinfo; cat /proc/loadavg; free -g | head -2; tools/_bin/pcietest; } > $OUT/box.txt 2>&1
timeout 1500 python3 -m pytest tests -m gpu -q --deselect tests/test_gpu_dropin_cli.py::test_block_loop_through_the_shim_reaches_the_batch_decode_rate 2>&1 | tail -5 > $OUT/pytest.log
cat /proc/loadavg >> $OUT/box.txt
for i in 1 2 3; do timeout 600 python3 -m pytest tests/test_gpu_dropin_cli.py -m gpu -q -s -k batch_decode_rate 2>&1 | grep -E "threads through|passed|failed" >> $OUT/pipeline3.txt; done
tools/_bin/pcietest >> $OUT/box.txt 2>&1
cat $OUT/box.txt $OUT/pytest.log $OUT/pipeline3.txt
