#!/usr/bin/env python3
"""per-round active / large-group counts + per-kernel breakdown of the forward BWT on one block
   python tools/sa_rounds.py [bytes] [kind ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')
import torch

import jampack_amd as jam

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64 << 20
kinds = sys.argv[2:] or ["text", "text_survey"]
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream()
ctx = jam.Context(0, st.cuda_stream)
for kind in kinds:
    t = jam.corpus.make(kind, n, 8)
    d_in = torch.from_numpy(t).to(dev)
    d_bwt = torch.empty(n + 480, dtype=torch.uint8, device=dev)
    ctx.bwt_forward(d_in, n, d_bwt, n + 480)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(3):
        ctx.bwt_forward(d_in, n, d_bwt, n + 480)
    e1.record(st)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    s = ctx.stats()
    print(f"== {kind} n={n}: {ms:.3f} ms = {n / ms / 1e6:.2f} GB/s, rounds={s.sa_rounds}")
    print("   active:", list(s.sa_round_active)[: s.sa_rounds])
    print("   large :", list(s.sa_round_large)[: s.sa_rounds])
    ctx.profile_enable(2)
    ctx.bwt_forward(d_in, n, d_bwt, n + 480)
    torch.cuda.synchronize()
    for row in sorted(ctx.profile_table(), key=lambda r: -r["ms"]):
        print(f"   {row['name']:38s} {row['ms']:9.3f} ms  launches={row['launches']:5d}  units={row['units']}")
    ctx.profile_enable(0)
