#!/usr/bin/env python3
"""soak: N random blocks (mixed generators and slices of real files, ragged sizes) through compress -> batch decompress; small ones also against the oracle.
   python tools/soak.py [N=300] [seed=1]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import jampack_amd as jam
from oracle.pyoracle import Oracle, Ref

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
o = Oracle()
ref = Ref() if Ref.available() else None          # the real reference (oracle/_ref): every block is compared with it, whatever its size
dev = torch.device("cuda", 0)
ctx = jam.Context(0, None)
kinds = ["text_survey", "text", "text_wide", "random", "runs", "geometric", "dna", "silesia", "zero", "two", "samples16", "repeat", "repeat4k"]
kinds = [k for k in kinds if k in jam.corpus.KINDS] if hasattr(jam.corpus, "KINDS") else kinds
# REAL bytes as well (round 6): random slices of this image's source trees and shared libraries (corpus.system_sources / system_binaries)
real = [x for x in (jam.corpus.system_sources(96 << 20), jam.corpus.system_binaries(96 << 20), jam.corpus.system_binaries(64 << 20, 200 << 20)) if x is not None]
t0 = time.time()
blocks, comp = [], []
checked = 0
for i in range(N):
    k = kinds[int(rng.integers(len(kinds)))]
    r = rng.random()
    n = int(rng.integers(1, 3000)) if r < 0.15 else (int(rng.integers(3000, 1_300_000)) if r < 0.7 else int(rng.integers(1_300_000, 9_000_000)))
    if real and rng.random() < 0.25:
        src = real[int(rng.integers(len(real)))]
        lo = int(rng.integers(0, len(src) - n))
        t = np.ascontiguousarray(src[lo:lo + n])
        k = "real"
    elif rng.random() < 0.35:
        # a phrase text over a random alphabet of 1..256 byte values (round 0's keys pack by the alphabet: every code width, real repeats)
        sigma = int(rng.choice([1, 2, 3, 4, 5, 8, 9, 16, 17, 28, 32, 33, 64, 65, 128, 129, 200, 256]))
        sym = np.sort(rng.choice(256, sigma, replace=False)).astype(np.uint8)
        book = [sym[rng.integers(0, sigma, int(rng.integers(3, 60)))] for _ in range(int(rng.integers(2, 300)))]
        parts, have = [], 0
        while have < n:
            ph = book[int(rng.integers(len(book)))]
            parts.append(ph); have += len(ph)
        t = np.ascontiguousarray(np.concatenate(parts)[:n])
        k = f"phrases{sigma}"
    else:
        try:
            t = jam.corpus.make(k, n, 10_000 * seed + i)
        except Exception:
            t = jam.corpus.make("text_survey", n, 10_000 * seed + i)
    d_t = torch.from_numpy(t).to(dev)
    cap = jam.ans_capacity(len(t) + jam.TRAILER)
    d_c = torch.empty(cap, dtype=torch.uint8, device=dev)
    m = ctx.block_compress(d_t, len(t), d_c, cap)
    c = d_c[:m].clone()
    if n < 400_000:
        want = o.ans_encode(o.bwt_forward(t))
        assert np.array_equal(c.cpu().numpy(), want), (i, k, n)
        checked += 1
    elif ref is not None and len(t) >= 120:
        want = ref.ans_encode(ref.bwt_forward(t))
        assert np.array_equal(c.cpu().numpy(), want), (i, k, n, "reference")
        checked += 1
    blocks.append(t); comp.append(c)
outs = [torch.empty(max(len(t), 1), dtype=torch.uint8, device=dev) for t in blocks]
for lo in range(0, N, 64):
    hi = min(N, lo + 64)
    n_, st = ctx.blocks_decompress(comp[lo:hi], [int(c.numel()) for c in comp[lo:hi]], outs[lo:hi], [len(t) for t in blocks[lo:hi]])
    for j in range(lo, hi):
        assert st[j - lo] == 0 and n_[j - lo] == len(blocks[j]) and np.array_equal(outs[j][: len(blocks[j])].cpu().numpy(), blocks[j]), (j, st[j - lo])
# and one at a time for a sample
for j in range(0, N, 7):
    one = torch.empty(max(len(blocks[j]), 1), dtype=torch.uint8, device=dev)
    assert ctx.block_decompress(comp[j], int(comp[j].numel()), one, len(blocks[j])) == len(blocks[j]) and np.array_equal(one[: len(blocks[j])].cpu().numpy(), blocks[j]), j
# the library's blocks-in-flight loop over everything, twice (launch groups and their streams come and go with the load)
for rep, nfl in enumerate((6, 3)):
    caps = [jam.ans_capacity(len(t) + jam.TRAILER) for t in blocks]
    d_ins = [torch.from_numpy(t).to(dev) for t in blocks]
    d_outs = [torch.empty(c, dtype=torch.uint8, device=dev) for c in caps]
    n_, st = ctx.blocks_compress(d_ins, [len(t) for t in blocks], d_outs, caps, nfl)
    for j in range(N):
        assert st[j] == 0 and n_[j] == comp[j].numel() and torch.equal(d_outs[j][: n_[j]], comp[j]), (rep, j)
    del d_ins, d_outs
print(f"soak ok: {N} blocks, {sum(len(b) for b in blocks) / 1e6:.0f} MB, {checked} compared with the oracle or the reference build byte for byte, {time.time() - t0:.0f} s")
