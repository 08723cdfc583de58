"""Round 0 of the suffix sort packs the text's alphabet (bwt_fwd.hip k_key_plan / k_pack_keys): sigma byte values -> ceil(log2 sigma)
bits each, 56 / bits bytes per key (7 for more than 128 values, 8 / 9 / 11 / 14 / 18 / 28 / 56 below).  The forward BWT must be the
reference's (divsufsort.cpp:1721 order, bwt.cpp:22-65 image) for every code width, for alphabets that are not contiguous, for runs and
periods measured against each width's own depth, at the end of the text and in the group sort of several small blocks.  -m gpu"""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def jam():
    import jampack_amd
    return jampack_amd

# (bits, depth) of the plan for an alphabet of `sigma` byte values
WIDTHS = {1: (1, 56), 2: (1, 56), 3: (2, 28), 4: (2, 28), 5: (3, 18), 8: (3, 18), 9: (4, 14), 16: (4, 14), 17: (5, 11), 28: (5, 11), 32: (5, 11),
          33: (6, 9), 64: (6, 9), 65: (7, 8), 128: (7, 8), 129: (8, 7), 200: (8, 7), 256: (8, 7)}


def _symbols(sigma, rng):
    """sigma distinct byte values that always include 0 and 255 when sigma >= 2 (codes are ranks, not byte values)"""
    if sigma == 1:
        return np.array([77], np.uint8)
    inner = rng.choice(np.arange(1, 255), sigma - 2, replace=False) if sigma > 2 else np.array([], np.int64)
    return np.sort(np.concatenate(([0, 255], inner))).astype(np.uint8)


def _text(sigma, n, seed):
    """random over the alphabet, with stretches that repeat at several distances, runs of one symbol around every depth and a tail
    that ends inside a run -- so that every width sees groups that survive round 0, run members and short suffixes"""
    rng = np.random.default_rng(seed)
    sym = _symbols(sigma, rng)
    t = sym[rng.integers(0, sigma, n)]
    pos = 0
    for L in (3, 7, 8, 11, 14, 18, 27, 28, 29, 55, 56, 57, 111, 112, 113, 300):
        if pos + 2 * L + 40 >= n:
            break
        t[pos: pos + L] = sym[rng.integers(0, sigma)]                 # a run
        pos += L + 5
        seg = t[pos: pos + L].copy()
        t[pos + L + 3: pos + 2 * L + 3] = seg                          # a repeat at distance L + 3
        pos += 2 * L + 9
    if n > 400:
        t[n - 130:] = sym[-1]                                          # the text ends inside a run of the largest symbol
        t[n - 300: n - 200] = np.tile(t[n - 300: n - 290], 10)         # period 10
    return np.ascontiguousarray(t)


@pytest.mark.parametrize("sigma", sorted(WIDTHS))
def test_forward_bwt_equals_oracle_for_every_code_width(jam, oracle, sigma):
    for n, seed in ((1, 1), (55, 2), (56, 3), (57, 4), (130, 5), (4097, 6), (70_001, 7)):
        t = _text(sigma, n, seed * 100 + sigma)
        out = np.full(n + 480, 0x11, dtype=np.uint8)
        got = jam.Bwt().ForwardBwt(t, out=out)
        exp = oracle.bwt_forward(t, prefill=0x11)
        assert np.array_equal(got, exp), f"sigma {sigma} n {n}"


@pytest.mark.parametrize("sigma", [4, 16, 28, 64, 128, 256])
def test_rounds_start_at_the_packed_depth(jam, sigma):
    """random text with planted repeats of exactly 10 bytes (the bytes in front of and behind the two copies differ): keys of 11 or more
    bytes (sigma <= 32) tell all suffixes apart in round 0, keys of 7..9 bytes leave the repeats to round 1"""
    import torch
    n = 1 << 20
    rng = np.random.default_rng(sigma)
    sym = _symbols(sigma, rng)
    t = sym[rng.integers(0, sigma, n)]
    for k in range(200):
        a, b = 1000 + 5000 * k, 1000 + 5000 * k + 2500
        t[b: b + 10] = t[a: a + 10]
        t[a - 1], t[b - 1] = sym[0], sym[1]
        t[a + 10], t[b + 10] = sym[0], sym[1]
    dev = torch.device("cuda", 0)
    ctx = jam.Context(0, torch.cuda.current_stream().cuda_stream)
    d_in = torch.from_numpy(np.ascontiguousarray(t)).to(dev)
    d_out = torch.empty(n + jam.TRAILER, dtype=torch.uint8, device=dev)
    ctx.bwt_forward(d_in, n, d_out, n + jam.TRAILER)
    torch.cuda.synchronize()
    s = ctx.stats()
    bits, depth = WIDTHS[sigma]
    assert s.sa_rounds == (1 if depth > 10 else 2), (sigma, depth, s.sa_rounds, list(s.sa_round_active[:4]))
    ctx.close()


def test_group_sort_of_blocks_with_different_alphabets(jam, oracle):
    """one group = one alphabet (the union): blocks over 2, 4, 28 and 200 symbols beside each other, every block's bytes the reference's"""
    import torch
    dev = torch.device("cuda", 0)
    ctx = jam.Context(0, torch.cuda.current_stream().cuda_stream)
    blocks = [_text(s, n, 9 * s + n) for s, n in ((2, 70_000), (4, 130_001), (28, 300_000), (200, 65_537), (1, 5_000), (28, 119), (3, 121))]
    d_in = [torch.from_numpy(b).to(dev) for b in blocks]
    caps = [jam.ans_capacity(len(b) + jam.TRAILER) for b in blocks]
    d_out = [torch.empty(c, dtype=torch.uint8, device=dev) for c in caps]
    n, st = ctx.blocks_compress(d_in, [len(b) for b in blocks], d_out, caps, 2)
    assert st == [0] * len(blocks)
    for i, b in enumerate(blocks):
        want = oracle.ans_encode(oracle.bwt_forward(b) if len(b) >= 120 else oracle.bwt_forward(b, prefill=0))
        assert n[i] == len(want) and np.array_equal(d_out[i][: n[i]].cpu().numpy(), want), i
    # a group whose blocks all use few symbols packs deep: the same bytes as one block at a time
    small = [_text(4, 200_000 + 7 * k, 50 + k) for k in range(6)]
    d_in = [torch.from_numpy(b).to(dev) for b in small]
    caps = [jam.ans_capacity(len(b) + jam.TRAILER) for b in small]
    d_out = [torch.empty(c, dtype=torch.uint8, device=dev) for c in caps]
    n, st = ctx.blocks_compress(d_in, [len(b) for b in small], d_out, caps, 2)
    assert st == [0] * len(small)
    for i, b in enumerate(small):
        one = torch.empty(caps[i], dtype=torch.uint8, device=dev)
        m = ctx.block_compress(d_in[i], len(b), one, caps[i])
        assert m == n[i] and torch.equal(one[:m], d_out[i][:m]), i
    ctx.close()


_CHILD = r"""
import sys, numpy as np
sys.path.insert(0, %r)
import jampack_amd as jam
from oracle.pyoracle import Oracle
o = Oracle()
ok = True
for kind, n in (("text_survey", 300_000), ("dna", 70_000), ("zero", 50_000), ("runs", 200_000), ("repeat4k", 100_000)):
    t = jam.corpus.make(kind, n, 3)
    ok = ok and np.array_equal(jam.Bwt().ForwardBwt(t), o.bwt_forward(t))
print("PLAIN_KEYS_OK" if ok else "PLAIN_KEYS_BAD")
"""


def test_plain_byte_keys_remain_a_working_comparator():
    """JPK_KEY_BITS=8: one byte per symbol whatever the alphabet (the A/B switch of profiles/r04_packed_keys.txt)"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, JPK_KEY_BITS="8")
    r = subprocess.run([sys.executable, "-c", _CHILD % root], env=env, capture_output=True, text=True, timeout=600)
    assert "PLAIN_KEYS_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
