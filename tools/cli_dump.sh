#!/bin/bash
cd "$(dirname "$0")/.."
bash tools/cli_repeat.sh > /dev/null 2>&1
g=0; b=0
for i in $(seq 1 40); do
  rm -rf /tmp/rp/d; mkdir -p /tmp/rp/d
  JPK_SHIM_DUMP=/tmp/rp/d oracle/_ref/jampack_shim_diag c /tmp/rp/in.bin /tmp/rp/o.jam -b1 -t1 > /dev/null 2>&1
  if cmp -s /tmp/rp/ref.jam /tmp/rp/o.jam; then [ $g -eq 0 ] && { rm -rf /tmp/rp/good; cp -r /tmp/rp/d /tmp/rp/good; g=1; }; else [ $b -eq 0 ] && { rm -rf /tmp/rp/bad; cp -r /tmp/rp/d /tmp/rp/bad; b=1; }; fi
  [ $g -eq 1 ] && [ $b -eq 1 ] && break
done
python - <<'PY'
import numpy as np, glob, os
for f in sorted(glob.glob("/tmp/rp/good/*.bin")):
    g = np.fromfile(f, dtype=np.uint8); bf = f.replace("/good/", "/bad/")
    if not os.path.exists(bf): print("no bad dump"); break
    b = np.fromfile(bf, dtype=np.uint8)
    if len(g) != len(b) or not np.array_equal(g, b):
        m = min(len(g), len(b)); d = np.flatnonzero(g[:m] != b[:m])
        print(os.path.basename(f), "len", len(g), len(b), "ndiff", len(d), "first", d[:10], "last", d[-5:])
        for k in d[:6]:
            print("   at", k, "good", g[max(0,k-4):k+8].tolist(), "bad", b[max(0,k-4):k+8].tolist())
        break
    else:
        print(os.path.basename(f), "same")
PY
