#!/bin/bash
# repeat the stock-CLI-through-the-shim compress and compare with the stock CLI's archive (flakiness hunt)
cd "$(dirname "$0")/.."
mkdir -p /tmp/rp && python - <<'PY'
import sys
sys.path.insert(0, '.')
import numpy as np
from jampack_amd import corpus as c
n, seed = 3_300_000, 41
parts = [c.make("text", n // 2, seed), c.make("samples16", n // 4, seed + 1), c.make("runs", n // 8, seed + 2)]
parts.append(c.make("random", n - sum(len(p) for p in parts), seed + 3))
np.concatenate(parts).tofile("/tmp/rp/in.bin")
PY
oracle/_ref/jampack_ref c /tmp/rp/in.bin /tmp/rp/ref.jam -b1 -t1 > /dev/null
bad=0
for i in $(seq 1 12); do
  for fl in "-b1 -t1 -T" "-b1 -t1" "-b1 -t2" "-b1 -t4"; do
    oracle/_ref/jampack_shim c /tmp/rp/in.bin /tmp/rp/o.jam $fl > /dev/null
    if ! cmp -s /tmp/rp/ref.jam /tmp/rp/o.jam; then bad=$((bad+1)); echo "MISMATCH run $i flags $fl size $(stat -c %s /tmp/rp/o.jam)"; cp /tmp/rp/o.jam /tmp/rp/bad_$bad.jam; fi
  done
done
echo "mismatches: $bad of 48"
if [ $bad -gt 0 ]; then python - <<'PY'
import numpy as np, glob
def frames(a):
    o_, fr = 0, []
    while o_ + 15 <= len(a):
        cs = int(a[o_ + 7: o_ + 11].view("<i4")[0]); fr.append(a[o_: o_ + 15 + cs]); o_ += 15 + cs
    return fr
ref = frames(np.fromfile("/tmp/rp/ref.jam", dtype=np.uint8))
for f in sorted(glob.glob("/tmp/rp/bad_*.jam")):
    fr = frames(np.fromfile(f, dtype=np.uint8))
    for i, (a, b) in enumerate(zip(ref, fr)):
        if len(a) != len(b) or not np.array_equal(a, b):
            m = min(len(a), len(b)); d = np.flatnonzero(a[:m] != b[:m])
            print(f, "frame", i, "len", len(a), len(b), "first diffs", d[:4], "hdr", a[:15].tolist(), b[:15].tolist())
PY
fi
