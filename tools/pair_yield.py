#!/usr/bin/env python3
"""what the pair rounds of the suffix sort yield: for blocks that take them, the list before and after every pair round
(python tools/pair_yield.py; honours JPK_PAIR_*)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np, torch
import jampack_amd as jam
from jampack_amd import corpus
dev = torch.device("cuda", 0)
ctx = jam.Context(0, None)
n = 64 << 20
a, b = np.array([0], dtype=np.uint8), np.array([0, 1], dtype=np.uint8)
while len(b) < (32 << 20): a, b = b, np.concatenate([b, a])
half = corpus.make("text_survey", 16 << 20, 3)
cases = {
    "real sources": corpus.system_sources(n),
    "silesia 64 MiB": corpus.make("silesia", n, 3),
    "silesia 212 MB (config 5)": corpus.load_or_make("silesia")[0],
    "repeat (1 MiB period)": corpus.make("repeat", n, 3),
    "repeat4k": corpus.make("repeat4k", n, 3),
    "twin text 32 MiB": np.concatenate([half, half]),
    "fibonacci word 32 MiB": b[: 32 << 20].copy(),
    "phrase-book text": corpus.make("text", n, 3),
    "sawtooth 32 MiB": (np.arange(32 << 20) % 251).astype(np.uint8),
    "twin real sources 2 x 32 MiB": (lambda h: None if h is None else np.concatenate([h, h]))(corpus.system_sources(32 << 20)),
    "real binaries twice 2 x 32 MiB": (lambda h: None if h is None else np.concatenate([h, h]))(corpus.system_binaries(32 << 20, 1000 << 20)),
    "real binaries (libMIOpen, first 64 MiB)": corpus.system_binaries(n),
    "real binaries from 200 MiB": corpus.system_binaries(n, 200 << 20),
    "real binaries from 3000 MiB": corpus.system_binaries(n, 3000 << 20),
}
if os.environ.get("JPK_YIELD_ONLY"):
    cases = {k: v for k, v in cases.items() if any(w in k for w in os.environ["JPK_YIELD_ONLY"].split(","))}
for name, t in cases.items():
    if t is None:
        continue
    t = np.ascontiguousarray(t); m = len(t)
    d_in = torch.from_numpy(t).to(dev)
    d_bwt = torch.empty(m + 480, dtype=torch.uint8, device=dev)
    ctx.bwt_forward(d_in, m, d_bwt, m + 480); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2): ctx.bwt_forward(d_in, m, d_bwt, m + 480)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 2 * 1e3
    s = ctx.stats(); act = list(s.sa_round_active)[: s.sa_rounds]
    pr = [r for r in range(40) if (s.sa_pair_rounds >> r) & 1]
    ys = [f"round {r}: {act[r]} -> {act[r + 1] if r + 1 < len(act) else 0} ({100.0 * (act[r + 1] if r + 1 < len(act) else 0) / max(act[r], 1):.0f} % left; the round before left {100.0 * act[r] / max(act[r - 1], 1):.0f} %)" for r in pr]
    print(f"{name:28s} {ms:8.2f} ms, rounds {s.sa_rounds:2d}, pair rounds {pr}: " + "; ".join(ys), flush=True)
    del d_in, d_bwt
