#!/bin/bash
set -u
REPO=$PWD
OUT=$REPO/gpurun_out/r3d
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_primitives.py tests/test_gpu_stages.py tests/test_gpu_parity.py -q 2>&1 | tail -60 > "$OUT/pytest_parity.log"
tail -40 "$OUT/pytest_parity.log"
