#!/usr/bin/env python3
"""stage times on adversarial 64 MiB blocks (run on the GPU box)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np, torch
import jampack_amd as jam
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64 << 20
dev = torch.device("cuda", 0); st = torch.cuda.current_stream()
ctx = jam.Context(0, st.cuda_stream)
for kind in ("text_survey", "text", "silesia", "zero", "repeat", "random", "dna", "two", "geometric", "samples16", "runs"):
    t = jam.corpus.make(kind, n, 3)
    d_in = torch.from_numpy(t).to(dev); cap = jam.ans_capacity(n + 480)
    d_bwt = torch.empty(n + 480, dtype=torch.uint8, device=dev); d_enc = torch.empty(cap, dtype=torch.uint8, device=dev)
    d_dec = torch.empty(n + 480, dtype=torch.uint8, device=dev); d_back = torch.empty(n, dtype=torch.uint8, device=dev)
    res = []
    for rep in range(2):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        e[0].record(st); ctx.bwt_forward(d_in, n, d_bwt, n + 480)
        e[1].record(st); cl = ctx.ans_encode(d_bwt, n + 480, d_enc, cap)
        e[2].record(st); ctx.ans_decode(d_enc, cl, d_dec, n + 480)
        e[3].record(st); ctx.bwt_inverse(d_dec, n + 480, d_back, n)
        e[4].record(st); torch.cuda.synchronize()
        res = [e[k].elapsed_time(e[k + 1]) for k in range(4)]
    ok = bool(torch.equal(d_back, d_in))
    s = ctx.stats()
    print(f"{kind:10s} fwd {res[0]:8.1f} ms  enc {res[1]:8.1f}  dec {res[2]:8.1f}  inv {res[3]:7.1f}  ratio {cl / n:.3f} alphabet {len(np.unique(t)):3d} key depth {s.sa_key_depth:2d} rounds {s.sa_rounds:2d} (pair rounds {[r for r in range(64) if (s.sa_pair_rounds >> r) & 1]}) sorted {s.sa_sorted_elems / n:.1f}n ok={ok} active/n {[round(x / n, 3) for x in s.sa_round_active[:min(s.sa_rounds, 12)]]}", flush=True)
