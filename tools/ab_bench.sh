#!/bin/bash
# same-box A/B of two builds of the library on the bench line (timed region only) and the small-block streams:
#   bash tools/ab_bench.sh     (tools/_bin/libjampack_amd_prev.so = the build to compare with)
cp jampack_amd/libjampack_amd.so /tmp/new.so
for r in 1 2 3; do
  for v in prev new; do
    if [ $v = prev ]; then cp tools/_bin/libjampack_amd_prev.so jampack_amd/libjampack_amd.so; else cp /tmp/new.so jampack_amd/libjampack_amd.so; fi
    echo "== $v bench"; timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>&1 | grep -o '"value": [0-9.]*' | head -1
    if [ $r != 3 ]; then echo "== $v small"; timeout 300 python tools/small_blocks.py 1,8 8,16 2>&1 | grep -v amdgpu | tail -2; fi
  done
done
cp /tmp/new.so jampack_amd/libjampack_amd.so
