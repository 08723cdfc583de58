#!/bin/bash
# is the STOCK reference CLI itself deterministic on this host?  (no GPU code involved)
cd "$(dirname "$0")/.."
bash tools/cli_repeat.sh > /dev/null 2>&1
nproc
for fl in "-b1 -t1" "-b1 -t2"; do
  bad=0
  for i in $(seq 1 30); do
    oracle/_ref/jampack_ref c /tmp/rp/in.bin /tmp/rp/s.jam $fl > /dev/null
    cmp -s /tmp/rp/ref.jam /tmp/rp/s.jam || bad=$((bad+1))
  done
  echo "stock CLI, flags $fl: $bad of 30 runs differ from the first stock run"
  bad=0
  for i in $(seq 1 30); do
    OMP_NUM_THREADS=1 oracle/_ref/jampack_ref c /tmp/rp/in.bin /tmp/rp/s.jam $fl > /dev/null
    cmp -s /tmp/rp/ref.jam /tmp/rp/s.jam || bad=$((bad+1))
  done
  echo "stock CLI, OMP_NUM_THREADS=1, flags $fl: $bad of 30 differ"
done
bad=0
for i in $(seq 1 30); do
  OMP_NUM_THREADS=1 oracle/_ref/jampack_shim c /tmp/rp/in.bin /tmp/rp/s.jam -b1 -t1 > /dev/null
  cmp -s /tmp/rp/ref.jam /tmp/rp/s.jam || bad=$((bad+1))
done
echo "shim CLI, OMP_NUM_THREADS=1, -b1 -t1: $bad of 30 differ"
