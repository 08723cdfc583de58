#!/usr/bin/env python3
"""two forward BWTs of a 64 MiB text block (for rocprofv3 --kernel-trace timelines)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')
import torch

import jampack_amd as jam

n = 64 << 20
t = jam.corpus.make("text", n, 8)
dev = torch.device("cuda", 0)
ctx = jam.Context(0, None)
d_in = torch.from_numpy(t).to(dev)
d_bwt = torch.empty(n + 480, dtype=torch.uint8, device=dev)
for _ in range(2):
    ctx.bwt_forward(d_in, n, d_bwt, n + 480)
torch.cuda.synchronize()
s = ctx.stats()
print("rounds", s.sa_rounds)
