#!/usr/bin/env python3
"""cycles / ns per encoder step of the slowest rANS chunk (jpk_stats.enc_chain_*):  python tools/chain_stats.py [kind] [bytes]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')
import torch

import jampack_amd as jam

kind = sys.argv[1] if len(sys.argv) > 1 else "text_survey"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 64 << 20
t = jam.corpus.make(kind, n, 8)
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream()
ctx = jam.Context(0, st.cuda_stream)
d_in = torch.from_numpy(t).to(dev)
cap = jam.ans_capacity(n + 480)
d_bwt = torch.empty(n + 480, dtype=torch.uint8, device=dev)
d_enc = torch.empty(cap, dtype=torch.uint8, device=dev)
ctx.bwt_forward(d_in, n, d_bwt, n + 480)
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    ctx.ans_encode(d_bwt, n + 480, d_enc, cap)
    e1.record(st)
    torch.cuda.synchronize()
    s = ctx.stats()
    print(f"{kind}: encode {e0.elapsed_time(e1):.3f} ms; slowest chunk: {s.enc_chain_steps} steps, {s.enc_chain_cycles} cycles = {s.enc_chain_cycles / max(s.enc_chain_steps, 1):.1f} cyc/step, "
          f"{s.enc_chain_ns / 1e6:.3f} ms = {s.enc_chain_ns / max(s.enc_chain_steps, 1):.2f} ns/step -> clock {s.enc_chain_cycles / max(s.enc_chain_ns, 1):.3f} GHz")
