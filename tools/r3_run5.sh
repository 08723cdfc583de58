#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r3e
mkdir -p "$OUT"
for V in 0 1 2 3 4; do
  JPK_RANS_STEP=$V timeout 300 python3 tools/diag_r3.py 2>&1 | grep -v "^done" >> "$OUT/diag_steps.txt"
done
cat "$OUT/diag_steps.txt"
