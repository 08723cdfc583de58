#!/usr/bin/env python3
"""one-off: round trip of a single large block (default 512 MiB, text-like + binary mix) through the fused entry points"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')
import numpy as np
import torch

import jampack_amd as jam

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512 << 20
t0 = time.time()
part = jam.corpus.make("silesia", min(n, 128 << 20), 9)
t = np.concatenate([part] * ((n + len(part) - 1) // len(part)))[:n].copy()
t[:: 4097] ^= 0x55                                  # break the exact period between the copies
print(f"corpus {n} B in {time.time() - t0:.1f} s")
dev = torch.device("cuda", 0)
ctx = jam.Context(0, None)
d_in = torch.from_numpy(t).to(dev)
cap = jam.ans_capacity(n + 480)
d_enc = torch.empty(cap, dtype=torch.uint8, device=dev)
d_back = torch.empty(n, dtype=torch.uint8, device=dev)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.time()
    cl = ctx.block_compress(d_in, n, d_enc, cap)
    torch.cuda.synchronize(); t1 = time.time()
    bl = ctx.block_decompress(d_enc, cl, d_back, n)
    torch.cuda.synchronize(); t2 = time.time()
    print(f"rep {rep}: compress {1e3 * (t1 - t0):.0f} ms ({n / 1e6 / (t1 - t0):.0f} MB/s)  decompress {1e3 * (t2 - t1):.0f} ms ({n / 1e6 / (t2 - t1):.0f} MB/s)  ratio {cl / n:.3f}")
print("round trip ok:", bl == n and bool(torch.equal(d_back, d_in)), " arena", ctx.stats().workspace_bytes >> 20, "MiB  rounds", ctx.stats().sa_rounds)
