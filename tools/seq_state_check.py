#!/usr/bin/env python3
"""state-dependence hunt: sequences of host-buffer calls of different sizes on ONE thread (one pooled context), every
result against the oracle; then the stock CLI through the shim with -t1 vs -t2, frame by frame."""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import jampack_amd as jam
from oracle.pyoracle import Oracle

o = Oracle()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def mixed(n, seed):
    c = jam.corpus
    parts = [c.make("text", n // 2, seed), c.make("samples16", n // 4, seed + 1), c.make("runs", n // 8, seed + 2)]
    parts.append(c.make("random", n - sum(len(p) for p in parts), seed + 3))
    return np.concatenate(parts)


data = mixed(3_300_000, 41)
MiB = 1 << 20
seqs = [[(0, MiB), (MiB, MiB), (2 * MiB, MiB), (3 * MiB, len(data) - 3 * MiB)],
        [(3 * MiB, len(data) - 3 * MiB), (0, MiB)],
        [(0, MiB), (3 * MiB, len(data) - 3 * MiB)],
        [(0, 2 * MiB), (2 * MiB, 100_000), (0, 3_000_000), (5, 70_000), (MiB, MiB)]]
bad = 0
for rep in range(3):
    for si, seq in enumerate(seqs):
        for (off, n) in seq:
            t = data[off:off + n]
            bw = jam.Bwt().ForwardBwt(t)
            eb = o.bwt_forward(t)
            if not np.array_equal(bw, eb):
                d = np.flatnonzero(bw[:len(eb)] != eb[:len(bw)])
                print(f"rep {rep} seq {si} block ({off},{n}): BWT differs at {d[:5]} (n diffs {len(d)})"); bad += 1
            en = jam.Ans().Encode(eb)
            ee = o.ans_encode(eb)
            if not np.array_equal(en, ee):
                print(f"rep {rep} seq {si} block ({off},{n}): ANS differs len {len(en)} vs {len(ee)}"); bad += 1
            fc = jam.block_compress(t)
            if not np.array_equal(fc, ee):
                print(f"rep {rep} seq {si} block ({off},{n}): fused differs len {len(fc)} vs {len(ee)}"); bad += 1
print("sequence check: bad =", bad)

REF, SHIM = os.path.join(ROOT, "oracle/_ref/jampack_ref"), os.path.join(ROOT, "oracle/_ref/jampack_shim")
if os.path.exists(SHIM):
    os.makedirs("/tmp/sq", exist_ok=True)
    data.tofile("/tmp/sq/in.bin")
    outs = {}
    for name, exe, flags in (("ref", REF, ["-b1", "-t1"]), ("s1", SHIM, ["-b1", "-t1"]), ("s1b", SHIM, ["-b1", "-t1"]), ("s2", SHIM, ["-b1", "-t2"]), ("s3", SHIM, ["-b1", "-t3"])):
        subprocess.run([exe, "c", "/tmp/sq/in.bin", f"/tmp/sq/{name}.jam"] + flags, stdout=subprocess.DEVNULL, check=True)
        outs[name] = np.fromfile(f"/tmp/sq/{name}.jam", dtype=np.uint8)

    def frames(a):
        o_, fr = 0, []
        while o_ + 15 <= len(a):
            cs = int(a[o_ + 7: o_ + 11].view("<i4")[0])
            fr.append(a[o_: o_ + 15 + cs])
            o_ += 15 + cs
        return fr

    fr = {k: frames(v) for k, v in outs.items()}
    for k in ("s1", "s1b", "s2", "s3"):
        for i, (a, b) in enumerate(zip(fr["ref"], fr[k])):
            if len(a) != len(b) or not np.array_equal(a, b):
                d = np.flatnonzero(a[:min(len(a), len(b))] != b[:min(len(a), len(b))])
                print(f"{k}: frame {i} differs: len {len(a)} vs {len(b)}, first diff at {d[:3]}, header ref {a[:15].tolist()} ours {b[:15].tolist()}")
        print(k, "frames", len(fr[k]), "total", len(outs[k]), "ref", len(outs["ref"]))
