#!/usr/bin/env python3
"""throughput of one stage alone with N contexts in flight (forward BWT only / rANS encode only / both):
   python tools/stage_scaling.py [kind] -- shows which stage fills the machine"""
import os, sys, time, queue, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import jampack_amd as jam

n = 64 << 20
t = jam.corpus.make(sys.argv[1] if len(sys.argv) > 1 else "text_survey", n, 8)
dev = torch.device("cuda", 0)
d_in = torch.from_numpy(t).to(dev)
cap = jam.ans_capacity(n + 480)
ctx0 = jam.Context(0, None)
d_bwt = torch.empty(n + 480, dtype=torch.uint8, device=dev)
ctx0.bwt_forward(d_in, n, d_bwt, n + 480)
for mode in ("fwd", "enc", "both"):
    for N in (1, 2, 3, 4, 6):
        ctxs = [jam.Context(0, None) for _ in range(N)]
        outs = [torch.empty(cap, dtype=torch.uint8, device=dev) for _ in range(N)]
        reps = 6 * N

        def work(k):
            for _ in range(reps // N + 1):
                if mode == "fwd":
                    ctxs[k].bwt_forward(d_in, n, outs[k], n + 480)
                elif mode == "enc":
                    ctxs[k].ans_encode(d_bwt, n + 480, outs[k], cap)
                else:
                    ctxs[k].block_compress(d_in, n, outs[k], cap)

        for c in range(N):
            (ctxs[c].bwt_forward(d_in, n, outs[c], n + 480), ctxs[c].ans_encode(d_bwt, n + 480, outs[c], cap))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        th = [threading.Thread(target=work, args=(k,)) for k in range(N)]
        [x.start() for x in th]; [x.join() for x in th]
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        tot = N * (reps // N + 1)
        print(f"{mode}: {N} contexts: {dt / tot * 1e3:7.2f} ms per 64 MiB block -> {tot * n / 1e6 / dt:7.0f} MB/s", flush=True)
        for c in ctxs: c.close()
