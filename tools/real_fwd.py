#!/usr/bin/env python3
"""forward BWT of the first 64 MiB of this image's real source files (corpus.system_sources): ms per block, rounds, pair rounds;
honours the JPK_PAIR_* knobs:   python tools/real_fwd.py [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')
import torch
import jampack_amd as jam
from jampack_amd import corpus
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n = 64 << 20
t = corpus.system_sources(n)
dev = torch.device("cuda", 0)
ctx = jam.Context(0, None)
d_in = torch.from_numpy(t).to(dev)
d_bwt = torch.empty(n + 480, dtype=torch.uint8, device=dev)
ctx.bwt_forward(d_in, n, d_bwt, n + 480)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(reps):
    ctx.bwt_forward(d_in, n, d_bwt, n + 480)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / reps * 1e3
s = ctx.stats()
pm = s.sa_pair_rounds
print(f"{ms:.2f} ms per block, rounds {s.sa_rounds}, pair rounds {[r for r in range(64) if (pm >> r) & 1]}, unresolved {list(s.sa_round_active)[: s.sa_rounds]}, in large groups {list(s.sa_round_large)[: s.sa_rounds]}")
import hashlib
print("bwt sha", hashlib.sha256(d_bwt.cpu().numpy().tobytes()).hexdigest()[:16])
