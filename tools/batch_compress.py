#!/usr/bin/env python3
"""jpk_dev_blocks_compress (the library's own blocks-in-flight loop) on the default workload: MB/s by blocks in flight"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import jampack_amd as jam
from jampack_amd import corpus

dev = torch.device("cuda", 0)
data, _ = corpus.load_or_make("enwik8")
blocks = corpus.split_blocks(data, 64 << 20)
d_in = [torch.from_numpy(np.ascontiguousarray(b)).to(dev) for b in blocks]
caps = [jam.ans_capacity(len(b) + jam.TRAILER) for b in blocks]
npl = 16
l_in = [d_in[i] for _ in range(npl) for i in range(len(blocks))]
l_len = [len(blocks[i]) for _ in range(npl) for i in range(len(blocks))]
l_cap = [caps[i] for _ in range(npl) for i in range(len(blocks))]
l_out = [torch.empty(c, dtype=torch.uint8, device=dev) for c in l_cap]
ctx = jam.Context(0, None)
mb = sum(len(b) for b in blocks) / 1e6
for nfl in (1, 2, 3, 4, 6, 8):
    ctx.blocks_compress(l_in, l_len, l_out, l_cap, nfl)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n, st = ctx.blocks_compress(l_in, l_len, l_out, l_cap, nfl)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / npl
    assert st == [0] * len(l_in)
    print(f"in flight {nfl}: {dt * 1e3:.2f} ms per 100 MB pass -> {mb / dt:.0f} MB/s", flush=True)
