#!/bin/bash
cd "$(dirname "$0")/.."
bash tools/cli_repeat.sh > /dev/null 2>&1
for i in $(seq 1 24); do
  JPK_SHIM_GUARD=1101004 JPK_SHIM_TRACE=1 oracle/_ref/jampack_shim_diag c /tmp/rp/in.bin /tmp/rp/o.jam -b1 -t1 > /dev/null 2> /tmp/rp/g.txt
  if cmp -s /tmp/rp/ref.jam /tmp/rp/o.jam; then r=ok; else r=BAD; fi
  echo "run $i $r $(grep -c guard /tmp/rp/g.txt) guard lines"; grep guard /tmp/rp/g.txt | sort | uniq -c
done
