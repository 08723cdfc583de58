#!/usr/bin/env python3
"""wall time of rANS encode / block compress of one 64 MiB block for JPK_ENC_GROUPS = 1..4 (no profiling events)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')
import torch

import jampack_amd as jam

n = 64 << 20
t = jam.corpus.make("text", n, 8)
dev = torch.device("cuda", 0)
ctx = jam.Context(0, None)
d_in = torch.from_numpy(t).to(dev)
cap = jam.ans_capacity(n + 480)
d_bwt = torch.empty(n + 480, dtype=torch.uint8, device=dev)
d_enc = torch.empty(cap, dtype=torch.uint8, device=dev)
ctx.bwt_forward(d_in, n, d_bwt, n + 480)
for g in (1, 2, 3, 4, 1, 4):
    os.environ["JPK_ENC_GROUPS"] = str(g)
    ctx.ans_encode(d_bwt, n + 480, d_enc, cap)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        ctx.ans_encode(d_bwt, n + 480, d_enc, cap)
    torch.cuda.synchronize()
    print(f"groups={g}: ans_encode {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms")
