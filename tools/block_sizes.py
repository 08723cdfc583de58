#!/usr/bin/env python3
"""MB/s compress and decompress per block size (the metric BASELINE.json names), enwik-like text, inputs resident in HBM:
one block at a time on one context, and a 256 MiB stream cut into blocks of that size with 4 blocks in flight (compress) /
all blocks in one batch call (decompress)."""
import os, sys, time, queue, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import jampack_amd as jam

dev = torch.device("cuda", 0)
total = 256 << 20
data = jam.corpus.make("text_survey", total, 8)
d_all = torch.from_numpy(data).to(dev)
nctx = 4
ctxs = [jam.Context(0, None) for _ in range(nctx)]
print("block MiB | one at a time: compress MB/s  decompress MB/s | stream of 256 MiB: compress MB/s (4 in flight)  decompress MB/s (one batch call)")
for mib in [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "1,2,4,8,16,32,64,128".split(","))]:
    bs = mib << 20
    nb = total // bs
    ins = [d_all[i * bs:(i + 1) * bs] for i in range(nb)]
    cap = jam.ans_capacity(bs + jam.TRAILER)
    outs = [torch.empty(cap, dtype=torch.uint8, device=dev) for _ in range(nb)]
    szs = [0] * nb
    c0 = ctxs[0]
    reps = max(2, min(8, 64 // mib))
    szs[0] = c0.block_compress(ins[0], bs, outs[0], cap)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        c0.block_compress(ins[0], bs, outs[0], cap)
    torch.cuda.synchronize(); tc1 = (time.perf_counter() - t0) / reps
    back = torch.empty(bs, dtype=torch.uint8, device=dev)
    c0.block_decompress(outs[0], szs[0], back, bs)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        c0.block_decompress(outs[0], szs[0], back, bs)
    torch.cuda.synchronize(); td1 = (time.perf_counter() - t0) / reps
    assert torch.equal(back, ins[0])

    def run_stream():
        q = queue.Queue()
        for i in range(nb):
            q.put(i)

        def w(k):
            while True:
                try:
                    i = q.get_nowait()
                except queue.Empty:
                    return
                szs[i] = ctxs[k].block_compress(ins[i], bs, outs[i], cap)
        th = [threading.Thread(target=w, args=(k,)) for k in range(nctx)]
        [t.start() for t in th]; [t.join() for t in th]

    run_stream()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    run_stream()
    torch.cuda.synchronize(); tcs = time.perf_counter() - t0
    backs = [torch.empty(bs, dtype=torch.uint8, device=dev) for _ in range(nb)]
    c0.blocks_decompress(outs, szs, backs, [bs] * nb)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n_, st_ = c0.blocks_decompress(outs, szs, backs, [bs] * nb)
    torch.cuda.synchronize(); tds = time.perf_counter() - t0
    assert st_ == [0] * nb and all(torch.equal(backs[i], ins[i]) for i in range(nb))
    print(f"{mib:9d} | {bs / 1e6 / tc1:10.0f} ({tc1 * 1e3:6.1f} ms) {bs / 1e6 / td1:10.0f} ({td1 * 1e3:6.1f} ms) | {total / 1e6 / tcs:10.0f} {total / 1e6 / tds:10.0f}   ratio {sum(szs) / total:.4f}", flush=True)
    del outs, backs
