#!/usr/bin/env python3
"""The two stages of the compress path as SEPARATE loops beside each other: NF contexts do nothing but forward BWTs, NE contexts nothing
but rANS encodes of a resident image, for a fixed time; blocks per second of each.  Tells what the encode stage takes from the suffix sort
(and the other way round) when neither waits for the other:   python tools/mix_stages.py [seconds] [NF,NE ...]"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
import numpy as np, torch
import jampack_amd as jam

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
mixes = [tuple(int(x) for x in a.split(",")) for a in sys.argv[2:]] or [(3, 0), (0, 5), (3, 5), (4, 4), (3, 3), (2, 6), (3, 8)]
n = 64 << 20
t = jam.corpus.make("text_survey", n, 8)
dev = torch.device("cuda", 0)
d_in = torch.from_numpy(t).to(dev)
cap = jam.ans_capacity(n + 480)
ctx0 = jam.Context(0, None)
d_bwt = torch.empty(n + 480, dtype=torch.uint8, device=dev)
ctx0.bwt_forward(d_in, n, d_bwt, n + 480)
NMAX = max(a + b for a, b in mixes)
ctxs = [jam.Context(0, None) for _ in range(NMAX)]
outs = [torch.empty(cap, dtype=torch.uint8, device=dev) for _ in range(NMAX)]
for c in range(NMAX):
    ctxs[c].bwt_forward(d_in, n, outs[c], n + 480)
    ctxs[c].ans_encode(d_bwt, n + 480, outs[c], cap)
torch.cuda.synchronize()
for nf, ne in mixes:
    done = [0] * (nf + ne)
    stop = [False]

    def work(k):
        while not stop[0]:
            if k < nf:
                ctxs[k].bwt_forward(d_in, n, outs[k], n + 480)
            else:
                ctxs[k].ans_encode(d_bwt, n + 480, outs[k], cap)
            done[k] += 1

    th = [threading.Thread(target=work, args=(k,)) for k in range(nf + ne)]
    t0 = time.perf_counter()
    [x.start() for x in th]
    time.sleep(secs)
    stop[0] = True
    [x.join() for x in th]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    f, e = sum(done[:nf]), sum(done[nf:])
    line = f"{nf} sorting + {ne} encoding contexts:"
    if nf:
        line += f"  forward BWT {dt / f * 1e3:6.2f} ms per block ({f * n / 1e6 / dt:6.0f} MB/s)"
    if ne:
        line += f"  encode {dt / e * 1e3:6.2f} ms per block ({e * n / 1e6 / dt:6.0f} MB/s)"
    if nf and ne:
        line += f"   -> a pipeline of both: {min(f, e) * n / 1e6 / dt:6.0f} MB/s"
    print(line, flush=True)
