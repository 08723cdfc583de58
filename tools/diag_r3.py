#!/usr/bin/env python3
"""round-3 diagnosis of the encoder chain variants (JPK_RANS_STEP=0..4): bytes vs the oracle on two text inputs, repeated, and the
slowest chain's cycles per step"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import jampack_amd as jam
from oracle.pyoracle import Oracle
o = Oracle()
tag = "STEP=" + os.environ.get("JPK_RANS_STEP", "default")
dev = torch.device("cuda", 0)
ctx = jam.Context(0, None)
for kind, n, seed in (("text", 2_500_000, 24), ("text_survey", 9_000_000, 900), ("geometric", 3_000_000, 5)):
    t = jam.corpus.make(kind, n, seed)
    ob = o.bwt_forward(t)
    oe = o.ans_encode(ob)
    d_in = torch.from_numpy(ob).to(dev)
    cap = jam.ans_capacity(len(ob))
    d_out = torch.empty(cap, dtype=torch.uint8, device=dev)
    bad = 0
    for rep in range(6):
        m = ctx.ans_encode(d_in, len(ob), d_out, cap)
        ok = m == len(oe) and np.array_equal(d_out[:m].cpu().numpy(), oe)
        bad += not ok
    st = ctx.stats()
    print(f"[{tag}] {kind} {n}: {6 - bad}/6 runs equal the oracle; slowest chain {st.enc_chain_cycles / max(st.enc_chain_steps, 1):.1f} cycles/step ({st.enc_chain_steps} steps)", flush=True)
print("done")
