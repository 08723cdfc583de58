#!/usr/bin/env python3
"""forward BWT of the first 64 MiB of this image's real source files (corpus.system_sources): ms per block, rounds, pair rounds;
honours the JPK_PAIR_* knobs:   python tools/real_fwd.py [reps] [sources|binaries[:<skip MiB>]]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')
import torch
import jampack_amd as jam
from jampack_amd import corpus
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n = 64 << 20
kind = sys.argv[2] if len(sys.argv) > 2 else "sources"          # sources | binaries | binaries:<skip MiB>
t = corpus.system_sources(n) if kind == "sources" else corpus.system_binaries(n, (int(kind.split(":")[1]) << 20) if ":" in kind else 0)
dev = torch.device("cuda", 0)
ctx = jam.Context(0, None)
d_in = torch.from_numpy(t).to(dev)
d_bwt = torch.empty(n + 480, dtype=torch.uint8, device=dev)
ctx.bwt_forward(d_in, n, d_bwt, n + 480)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(reps):
    ctx.bwt_forward(d_in, n, d_bwt, n + 480)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / reps * 1e3
s = ctx.stats()
pm = s.sa_pair_rounds
print(f"{ms:.2f} ms per block, rounds {s.sa_rounds}, pair rounds {[r for r in range(64) if (pm >> r) & 1]}, unresolved {list(s.sa_round_active)[: s.sa_rounds]}, in large groups {list(s.sa_round_large)[: s.sa_rounds]}")
import hashlib
print("bwt sha", hashlib.sha256(d_bwt.cpu().numpy().tobytes()).hexdigest()[:16])
cap = jam.ans_capacity(n + 480); d_enc = torch.empty(cap, dtype=torch.uint8, device=dev); d_back = torch.empty(n, dtype=torch.uint8, device=dev)
torch.cuda.synchronize(); t0 = time.perf_counter(); cl = ctx.block_compress(d_in, n, d_enc, cap); torch.cuda.synchronize(); t1 = time.perf_counter()
bl = ctx.block_decompress(d_enc, cl, d_back, n); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"one block: compress {1e3 * (t1 - t0):.1f} ms, decompress {1e3 * (t2 - t1):.1f} ms, ratio {cl / n:.3f}, round trip {bool(bl == n and torch.equal(d_back, d_in))}")
