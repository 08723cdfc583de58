#!/usr/bin/env python3
"""N forward BWTs of a 64 MiB block (for rocprofv3 --kernel-trace --stats):  python tools/fwd_once.py [kind] [reps] [MiB]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')
import torch

import jampack_amd as jam

kind = sys.argv[1] if len(sys.argv) > 1 else "text_survey"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
n = (int(sys.argv[3]) if len(sys.argv) > 3 else 64) << 20
t = jam.corpus.make(kind, n, 8)
dev = torch.device("cuda", 0)
ctx = jam.Context(0, None)
d_in = torch.from_numpy(t).to(dev)
d_bwt = torch.empty(n + 480, dtype=torch.uint8, device=dev)
import time
ctx.bwt_forward(d_in, n, d_bwt, n + 480)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    ctx.bwt_forward(d_in, n, d_bwt, n + 480)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / reps * 1e3
s = ctx.stats()
print(f"{ms:.2f} ms per block, key depth {s.sa_key_depth} bytes, rounds {s.sa_rounds}, unresolved at the start of each {list(s.sa_round_active)[: s.sa_rounds]}")
