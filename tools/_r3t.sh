#!/bin/bash
export JAMPACK_CORPUS_CACHE=/tmp/jpk_corpus
mkdir -p gpurun_out/r3t
for L in 1 2 3 4; do echo "lanes $L"; JPK_INV_LANES=$L NLIST=16,64 python3 tools/dec_scaling.py batch 2>/dev/null; done | tee gpurun_out/r3t/lanes.txt
