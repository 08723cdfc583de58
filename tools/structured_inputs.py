#!/usr/bin/env python3
"""Structured inputs probed for cliffs of the suffix sort: forward BWT of a 32 MiB block of counters, fixed-width records, a sawtooth, the
Fibonacci word, base64, sparse data and a text that holds everything twice -- time, rounds, pair rounds, key code, and the image against
the reference build (oracle/_ref).   python tools/structured_inputs.py   -> profiles/r05_structured_inputs.txt"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import jampack_amd as jam
from oracle.pyoracle import Ref
ref = Ref()
dev = torch.device("cuda", 0)
ctx = jam.Context(0, None)
rng = np.random.default_rng(5)
n = 32 << 20
def records():
    rec = np.zeros((n // 32, 32), dtype=np.uint8)
    rec[:, :8] = np.frombuffer(np.arange(n // 32, dtype=np.uint64).tobytes(), dtype=np.uint8).reshape(-1, 8)
    rec[:, 8:12] = rng.integers(0, 256, (n // 32, 4))
    rec[:, 12:20] = np.frombuffer(b"CONSTANT", dtype=np.uint8)
    rec[:, 20:24] = rng.integers(48, 58, (n // 32, 4))
    return rec.ravel()
cases = {
    "counter32": np.frombuffer(np.arange(n // 4, dtype=np.uint32).tobytes(), dtype=np.uint8),
    "counter32be": np.frombuffer(np.arange(n // 4, dtype=">u4").tobytes(), dtype=np.uint8),
    "records32": records(),
    "sawtooth": (np.arange(n) % 251).astype(np.uint8),
    "fib-like": None,
    "base64": np.frombuffer(b"ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/", dtype=np.uint8)[rng.integers(0, 64, n)],
    "sparse": np.where(rng.random(n) < 0.02, rng.integers(1, 256, n), 0).astype(np.uint8),
}
a, b = np.array([0], dtype=np.uint8), np.array([0, 1], dtype=np.uint8)
while len(b) < n: a, b = b, np.concatenate([b, a])
cases["fib-like"] = b[:n].copy()
_half = jam.corpus.make("text_survey", n // 2, 3)
cases["twin-text"] = np.concatenate([_half, _half])
for name, t in cases.items():
    t = np.ascontiguousarray(t)
    d_in = torch.from_numpy(t.copy()).to(dev); m = len(t)
    d_bwt = torch.empty(m + 480, dtype=torch.uint8, device=dev)
    ctx.bwt_forward(d_in, m, d_bwt, m + 480); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2): ctx.bwt_forward(d_in, m, d_bwt, m + 480)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 2 * 1e3
    s = ctx.stats()
    ok = np.array_equal(d_bwt.cpu().numpy(), ref.bwt_forward(t, prefill=0)[: m + 480])
    print("%-12s %7.2f ms per 32 MiB, rounds %2d, pair rounds %s, key order %d depth %d, equal to the reference: %s" % (name, ms, s.sa_rounds, bin(s.sa_pair_rounds), s.sa_key_order, s.sa_key_depth, ok), flush=True)
