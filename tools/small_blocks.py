#!/usr/bin/env python3
"""The reference's default block size is 8 MiB (format.hpp:20): MB/s of a 256 MiB text stream cut into 1 / 8 / 64 MiB blocks through
ONE jpk_dev_blocks_compress / jpk_dev_blocks_decompress call, by blocks in flight.   python tools/small_blocks.py [sizes MiB] [in flight list]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import jampack_amd as jam

dev = torch.device("cuda", 0)
total = 256 << 20
data = jam.corpus.make("text_survey", total, 8)
d_all = torch.from_numpy(data).to(dev)
sizes = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1,8,64").split(",")]
flights = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "4,8,16,32").split(",")]
ctx = jam.Context(0, None)
for mib in sizes:
    bs = mib << 20
    nb = total // bs
    ins = [d_all[i * bs:(i + 1) * bs] for i in range(nb)]
    cap = jam.ans_capacity(bs + jam.TRAILER)
    outs = [torch.empty(cap, dtype=torch.uint8, device=dev) for _ in range(nb)]
    row = []
    for nfl in flights:
        if nfl > nb:
            continue
        ctx.blocks_compress(ins, [bs] * nb, outs, [cap] * nb, nfl)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n, st = ctx.blocks_compress(ins, [bs] * nb, outs, [cap] * nb, nfl)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        assert st == [0] * nb
        row.append(f"{nfl} in flight {total / 1e6 / dt:6.0f}")
    backs = [torch.empty(bs, dtype=torch.uint8, device=dev) for _ in range(nb)]
    ctx.blocks_decompress(outs, n, backs, [bs] * nb)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ln, st = ctx.blocks_decompress(outs, n, backs, [bs] * nb)
    torch.cuda.synchronize(); dd = time.perf_counter() - t0
    assert st == [0] * nb and all(torch.equal(b, i) for b, i in zip(backs, ins))
    print(f"{mib:3d} MiB blocks x {nb:3d}: compress MB/s: " + " | ".join(row) + f" || decompress (one call) {total / 1e6 / dd:6.0f} MB/s", flush=True)
    jam.shutdown()
    ctx = jam.Context(0, None)
