#!/usr/bin/env python3
"""per-kernel HIP-event breakdown of one stage on one block (run on the GPU box)
   python tools/prof_stage.py [fwd|inv|enc|dec|all] [bytes] [kind]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')
import numpy as np
import torch

import jampack_amd as jam

stage = sys.argv[1] if len(sys.argv) > 1 else "all"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 64 << 20
kind = sys.argv[3] if len(sys.argv) > 3 else "text"
t = jam.corpus.make(kind, n, 8)
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream()
ctx = jam.Context(0, st.cuda_stream)
d_in = torch.from_numpy(t).to(dev)
cap = jam.ans_capacity(n + 480)
d_bwt = torch.empty(n + 480, dtype=torch.uint8, device=dev)
d_enc = torch.empty(cap, dtype=torch.uint8, device=dev)
d_dec = torch.empty(n + 480, dtype=torch.uint8, device=dev)
d_back = torch.empty(n, dtype=torch.uint8, device=dev)


def run(name, fn, reps=2):
    fn()  # warm (arena growth)
    ctx.profile_enable(2)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps):
        r = fn()
    e1.record(st)
    torch.cuda.synchronize()
    tab = ctx.profile_table()
    ctx.profile_enable(0)
    tot = e0.elapsed_time(e1) / reps
    print(f"== {name}: {tot:.3f} ms/call ({n / 1e6 / tot * 1e3:.0f} MB/s)  [profiled: events add overhead]")
    for row in sorted(tab, key=lambda r: -r["ms"]):
        print(f"   {row['name']:38s} {row['ms'] / reps:9.3f} ms  launches/call={row['launches'] // reps:5d}  units/call={row['units'] // reps}")
    s = ctx.stats()
    print(f"   stats: sa_rounds={s.sa_rounds} sa_sorted_elems={s.sa_sorted_elems} splitters={s.inv_splitters} ovf={s.inv_overflow_slots} chunks={s.ans_chunks} rle={s.ans_rle_symbols}")
    return r


if stage in ("fwd", "all"):
    run("forward_bwt", lambda: ctx.bwt_forward(d_in, n, d_bwt, n + 480))
else:
    ctx.bwt_forward(d_in, n, d_bwt, n + 480)
clen = ctx.ans_encode(d_bwt, n + 480, d_enc, cap)
if stage in ("enc", "all"):
    run("ans_encode", lambda: ctx.ans_encode(d_bwt, n + 480, d_enc, cap))
if stage in ("dec", "all"):
    run("ans_decode", lambda: ctx.ans_decode(d_enc, clen, d_dec, n + 480), reps=1)
else:
    ctx.ans_decode(d_enc, clen, d_dec, n + 480)
if stage in ("inv", "all"):
    run("inverse_bwt", lambda: ctx.bwt_inverse(d_dec, n + 480, d_back, n))
    assert torch.equal(d_back, d_in)
