#!/usr/bin/env python3
"""per-chunk header fields (olen, clen, rlen) of the Ans stream of one 64 MiB block, and what the two serial decode kernels
therefore pay per RLE0 symbol in their slowest chunk (kernel times from the in-library HIP-event profiler)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import jampack_amd as jam

C = [0, 127, 16510, 2113661, 270549116]


def leb(buf, p):
    x = 0; k = 0
    while True:
        b = int(buf[p]); p += 1
        x = (x << 7) | (b & 0x7f)
        if b & 0x80:
            return x + (C[k] if k else 0), p
        k += 1


n = 64 << 20
t = jam.corpus.make(sys.argv[1] if len(sys.argv) > 1 else "text_survey", n, 8)
dev = torch.device("cuda", 0)
ctx = jam.Context(0, None)
d_in = torch.from_numpy(t).to(dev)
cap = jam.ans_capacity(n + 480)
d_enc = torch.empty(cap, dtype=torch.uint8, device=dev)
clen = ctx.block_compress(d_in, n, d_enc, cap)
enc = d_enc[:clen].cpu().numpy()
p = 0; rl = []; ol = []
while p < len(enc):
    vals = []
    for _ in range(259):
        v, p = leb(enc, p); vals.append(v)
    ol.append(vals[256]); rl.append(vals[258]); p += vals[257]
rl = np.array(rl)
print(f"{len(rl)} chunks: rlen mean {rl.mean():.0f} max {rl.max()} min {rl.min()}  (olen {ol[0]})")
d_out = torch.empty(n, dtype=torch.uint8, device=dev)
ctx.block_decompress(d_enc, clen, d_out, n)
ctx.profile_enable(2)
ctx.block_decompress(d_enc, clen, d_out, n)
for row in ctx.profile_table():
    nm = row[0] if isinstance(row, (tuple, list)) else row.get("name")
    print(row)
