#!/bin/bash
cd "$(dirname "$0")/.."
oracle/_ref/jampack_ref c /tmp/rp/in.bin /tmp/rp/ref.jam -b1 -t1 > /dev/null 2>&1 || { bash tools/cli_repeat.sh > /dev/null; }
for i in $(seq 1 16); do
  JPK_SHIM_TRACE=1 oracle/_ref/jampack_shim_diag c /tmp/rp/in.bin /tmp/rp/o.jam -b1 -t1 > /dev/null 2> /tmp/rp/trace_$i.txt
  if cmp -s /tmp/rp/ref.jam /tmp/rp/o.jam; then echo "run $i ok"; cp /tmp/rp/trace_$i.txt /tmp/rp/good.txt; else echo "run $i BAD"; cp /tmp/rp/trace_$i.txt /tmp/rp/bad.txt; fi
done
echo "--- good"; cat /tmp/rp/good.txt; echo "--- bad"; cat /tmp/rp/bad.txt 2>/dev/null
