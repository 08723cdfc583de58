#!/bin/bash
# same-box A/B of two builds of the library on the bench line (timed region only) and the small-block streams:
#   bash tools/ab_bench.sh     (tools/_bin/libjampack_amd_prev.so = the build to compare with)
# The build is chosen per process through JPK_LIB (jampack_amd/_lib.py): the product library is never replaced in place.
set -u
prev=tools/_bin/libjampack_amd_prev.so
[ -f "$prev" ] || { echo "missing $prev" >&2; exit 1; }
for r in 1 2 3; do
  for v in prev new; do
    if [ $v = prev ]; then lib=$PWD/$prev; else lib=$PWD/jampack_amd/libjampack_amd.so; fi
    echo "== $v bench"; JPK_LIB=$lib timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>&1 | grep -o '"value": [0-9.]*' | head -1
    if [ $r != 3 ]; then echo "== $v small"; JPK_LIB=$lib timeout 300 python tools/small_blocks.py 1,8 8,16 2>&1 | grep -v amdgpu | tail -2; fi
  done
done
