#!/usr/bin/env python3
"""N rANS decodes of one 64 MiB block's stream (for rocprofv3 kernel traces / SQ counters of k_dec_rans and k_dec_rank):
   python tools/dec_once.py [kind] [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')
import torch

import jampack_amd as jam

kind = sys.argv[1] if len(sys.argv) > 1 else "text_survey"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
n = 64 << 20
t = jam.corpus.make(kind, n, 8)
dev = torch.device("cuda", 0)
ctx = jam.Context(0, None)
d_in = torch.from_numpy(t).to(dev)
cap = jam.ans_capacity(n + 480)
d_bwt = torch.empty(n + 480, dtype=torch.uint8, device=dev)
d_enc = torch.empty(cap, dtype=torch.uint8, device=dev)
d_dec = torch.empty(n + 480, dtype=torch.uint8, device=dev)
ctx.bwt_forward(d_in, n, d_bwt, n + 480)
clen = ctx.ans_encode(d_bwt, n + 480, d_enc, cap)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(reps):
    dl = ctx.ans_decode(d_enc, clen, d_dec, n + 480)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / reps * 1e3
s = ctx.stats()
assert dl == n + 480 and torch.equal(d_dec, d_bwt)
print(f"{ms:.1f} ms per decode, {s.ans_chunks} chunks, {s.ans_rle_symbols} RLE0 symbols ({s.ans_rle_symbols / (n + 480):.3f} per byte), {reps} decode(s)")
