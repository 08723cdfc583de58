#!/usr/bin/env python3
"""decode latency vs blocks in flight: N contexts decode the same 64 MiB block concurrently"""
import os, sys, time
import concurrent.futures as cf
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')
import numpy as np, torch
import jampack_amd as jam

n = 64 << 20
t = jam.corpus.make(sys.argv[2] if len(sys.argv) > 2 else "text_survey", n, 8)
dev = torch.device("cuda", 0)
ctx0 = jam.Context(0, None)
d_in = torch.from_numpy(t).to(dev)
cap = jam.ans_capacity(n + 480)
d_enc = torch.empty(cap, dtype=torch.uint8, device=dev)
clen = ctx0.block_compress(d_in, n, d_enc, cap)
d_bwt = torch.empty(n + 480, dtype=torch.uint8, device=dev)
ctx0.bwt_forward(d_in, n, d_bwt, n + 480)
mode = sys.argv[1] if len(sys.argv) > 1 else "ans"
if mode == "batch":
    for N in [int(x) for x in os.environ.get("NLIST", "1,4,8,16,24,32").split(",")]:
        outs = [torch.empty(n, dtype=torch.uint8, device=dev) for _ in range(N)]
        encs = [d_enc[:clen].clone() for _ in range(N)]
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            ln, st = ctx0.blocks_decompress(encs, [clen] * N, outs, [n] * N)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        assert st == [0] * N and all(torch.equal(o, d_in) for o in outs)
        print(f"batch: {N} blocks in one call: {dt * 1e3:.1f} ms -> {N * n / 1e6 / dt:.0f} MB/s")
    sys.exit(0)
for N in (1, 4, 8, 16, 24):
    ctxs = [jam.Context(0, None) for _ in range(N)]
    outs = [torch.empty(n + 480, dtype=torch.uint8, device=dev) for _ in range(N)]
    encs = [d_enc[:clen].clone() for _ in range(N)]
    pool = cf.ThreadPoolExecutor(max_workers=N)

    def work(k):
        if mode == "ans":
            ctxs[k].ans_decode(encs[k], clen, outs[k], n + 480)
        elif mode == "block":
            ctxs[k].block_decompress(encs[k], clen, outs[k], n)
        else:
            ctxs[k].ans_encode(d_bwt, n + 480, outs[k], n + 480)

    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        list(pool.map(work, range(N)))
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{mode}: {N} in flight: {dt * 1e3:.1f} ms for {N} blocks -> {N * n / 1e6 / dt:.0f} MB/s")
    pool.shutdown()
    for c in ctxs: c.close()
