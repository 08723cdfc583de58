#!/usr/bin/env python3
"""one forward BWT + three rANS encodes of a 64 MiB text block (for rocprofv3 --kernel-trace timelines)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')
import torch

import jampack_amd as jam

n = 64 << 20
t = jam.corpus.make(sys.argv[1] if len(sys.argv) > 1 else "text_survey", n, 8)
dev = torch.device("cuda", 0)
ctx = jam.Context(0, None)
d_in = torch.from_numpy(t).to(dev)
cap = jam.ans_capacity(n + 480)
d_bwt = torch.empty(n + 480, dtype=torch.uint8, device=dev)
d_enc = torch.empty(cap, dtype=torch.uint8, device=dev)
ctx.bwt_forward(d_in, n, d_bwt, n + 480)
for _ in range(3):
    ctx.ans_encode(d_bwt, n + 480, d_enc, cap)
torch.cuda.synchronize()
