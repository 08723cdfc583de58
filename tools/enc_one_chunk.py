#!/usr/bin/env python3
"""per-kernel time of the entropy encode stage on the single densest 1 MiB chunk of the 64 MiB text block"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')
import numpy as np
import torch

import jampack_amd as jam

n = 64 << 20
t = jam.corpus.make("text", n, 8)
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream()
ctx = jam.Context(0, st.cuda_stream)
d_in = torch.from_numpy(t).to(dev)
d_bwt = torch.empty(n + 480, dtype=torch.uint8, device=dev)
ctx.bwt_forward(d_in, n, d_bwt, n + 480)
bwt = d_bwt.cpu().numpy()
# densest chunk = fewest zero-rank candidates ~ most byte changes
chg = [(int((bwt[c + 1:c + (1 << 20)] != bwt[c:c + (1 << 20) - 1]).sum()), c) for c in range(0, n, 1 << 20)]
chg.sort(reverse=True)
print("most changes per chunk:", chg[:3])
c0 = chg[0][1]
one = torch.from_numpy(bwt[c0:c0 + (1 << 20)].copy()).to(dev)
cap = jam.ans_capacity(1 << 20)
d_enc = torch.empty(cap, dtype=torch.uint8, device=dev)
ctx.ans_encode(one, 1 << 20, d_enc, cap)
ctx.profile_enable(2)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(st)
for _ in range(3):
    ctx.ans_encode(one, 1 << 20, d_enc, cap)
e1.record(st)
torch.cuda.synchronize()
print(f"one chunk: {e0.elapsed_time(e1) / 3:.3f} ms/call")
for r in sorted(ctx.profile_table(), key=lambda r: -r["ms"]):
    print(f"   {r['name']:40s} {r['ms'] / 3:8.3f} ms  launches/call={r['launches'] // 3}")
print("rle symbols:", ctx.stats().ans_rle_symbols)
