"""Group-size distribution of the suffix sort's doubling rounds (CPU, numpy): for h = 7, 14, 28 ... the histogram of the sizes
of the groups of suffixes that still tie on their first h bytes, weighted by members.  Tells which share of k_seg_round's
work sits in groups small enough for an all-pairs ranking instead of the LDS radix sort.
usage: python tools/group_sizes.py [workload] [nbytes]"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from jampack_amd import corpus

name = sys.argv[1] if len(sys.argv) > 1 else "enwik8"
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 64 << 20
T, _ = corpus.load_or_make(name, count=nb)
n = len(T) - len(T) % 120
T = T[:n]
pad = np.concatenate((T, np.zeros(8, np.uint8)))
key = np.zeros(n, np.uint64)
for k in range(7):
    key |= pad[k:k + n].astype(np.uint64) << np.uint64(8 * (6 - k))
# suffixes shorter than 7 bytes: treat as unique by construction (negligible)
t0 = time.time()
order = np.argsort(key, kind="stable")
ks = key[order]
head = np.ones(n, bool); head[1:] = ks[1:] != ks[:-1]
rank = np.empty(n, np.int64)
grp_of_pos = np.maximum.accumulate(np.where(head, np.arange(n), 0))
rank[order] = grp_of_pos
print("round0 sort %.1fs" % (time.time() - t0), flush=True)
bins = [1, 2, 3, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 4096, 1 << 16, 1 << 20, 1 << 30]
h = 7
while True:
    starts = np.flatnonzero(head)
    sizes = np.diff(np.concatenate((starts, [n])))
    active = sizes[sizes > 1]
    tot = int(active.sum())
    print(f"h={h}: groups>1: {len(active)}  members: {tot} ({100.0*tot/n:.1f}% of n)  sum g^2={float((active.astype(np.float64)**2).sum()):.3e}")
    if tot == 0:
        break
    cum = 0
    for lo, hi in zip(bins[1:-1], bins[2:]):
        sel = active[(active >= lo) & (active < hi)]
        s = int(sel.sum()); cum += s
        print(f"   size [{lo},{hi}): groups {len(sel):9d} members {s:10d} {100.0*s/tot:5.1f}%  cum {100.0*cum/tot:5.1f}%  sum g^2/members={float((sel.astype(np.float64)**2).sum())/max(s,1):.1f}")
    # next round: sort by (rank, rank[sa+h])
    sa = order
    k2 = np.where(sa + h < n, rank[np.minimum(sa + h, n - 1)] + 1, 0)
    comp = (grp_of_pos.astype(np.uint64) << np.uint64(32)) | k2.astype(np.uint64)
    o2 = np.argsort(comp, kind="stable")
    order = sa[o2]
    cs = comp[o2]
    head = np.ones(n, bool); head[1:] = cs[1:] != cs[:-1]
    grp_of_pos = np.maximum.accumulate(np.where(head, np.arange(n), 0))
    rank[order] = grp_of_pos
    h *= 2
    print("round done %.1fs" % (time.time() - t0), flush=True)
