"""The pair rule of the suffix sort on the GPU (bwt_fwd.hip k_pair_*): long repeats are resolved by induction from their successors
(what divsufsort.cpp:1427-1520 does by induced sorting) instead of log2(LCP) doubling rounds.  The forward BWT must stay the
reference's (bwt.cpp:22-65) -- on tiny inputs with the rule forced on from the third round (child processes: the knobs are read once),
on window / tile boundaries, in the group sort of several blocks, with the rule off (comparator), and at 8-16 MiB with the default
knobs, where the repeated-segment blocks must finish in a few rounds.  -m gpu"""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_CHILD = r"""
import sys, random, numpy as np
sys.path.insert(0, %(root)r)
sys.path.insert(0, %(root)r + "/tests")
import jampack_amd as jam
from oracle.pyoracle import Oracle
from test_pair_rule_model import _gen
o = Oracle()
rng = random.Random(%(seed)d)
bad = 0
used = 0
def check(t, what):
    global bad, used
    t = np.frombuffer(bytes(t), dtype=np.uint8)
    got = jam.Bwt().ForwardBwt(t)
    exp = o.bwt_forward(t)
    if not np.array_equal(got, exp):
        bad += 1
        print("MISMATCH", what, len(t))
# 1. the model test's generators at GPU-test sizes: every structure the rule was derived for
for it in range(%(iters)d):
    base = _gen(rng)
    reps = rng.choice([1, 1, 2, 5, 30])
    t = (base * reps) + bytes(rng.randrange(3) for _ in range(rng.randrange(4)))
    if len(t) < 120:
        t = t + bytes(rng.randrange(2) for _ in range(130))
    check(t, ("gen", it))
# 2. stretches that cross the 1024-slot windows and the 4096-position tiles of the kernels: a segment repeated with periods around them
nrng = np.random.default_rng(%(seed)d)
for period in (1, 2, 3, 63, 64, 65, 1023, 1024, 1025, 4095, 4096, 4097, 10_000):
    for sigma in (2, 4, 200):
        seg = nrng.integers(0, sigma, period).astype(np.uint8)
        for total in (period * 3 + 5, 40_000 + period):
            t = np.concatenate((nrng.integers(0, sigma, 500).astype(np.uint8), np.tile(seg, total // period + 1)[:total], nrng.integers(0, sigma, 700).astype(np.uint8)))
            check(t.tobytes(), ("period", period, sigma, total))
# 3. two and three copies of long segments at odd distances, the last copy ending the text (verdict by the empty suffix)
for L in (100, 5_000, 70_000):
    for sigma in (3, 26, 256):
        seg = nrng.integers(0, sigma, L).astype(np.uint8)
        f = lambda k: nrng.integers(0, sigma, k).astype(np.uint8)
        check(np.concatenate((f(300), seg, f(1234), seg, f(77))).tobytes(), ("two", L, sigma))
        check(np.concatenate((seg, f(5), seg, f(900), seg)).tobytes(), ("three-end", L, sigma))
        check(np.concatenate((f(130), seg, seg[: L // 2])).tobytes(), ("ends-inside", L, sigma))
print("PAIR_OK" if bad == 0 else "PAIR_BAD %%d" %% bad)
"""


def _run_child(env_extra, iters=250, seed=11, timeout=1500):
    env = dict(os.environ, **env_extra)
    r = subprocess.run([sys.executable, "-c", _CHILD % {"root": ROOT, "iters": iters, "seed": seed}], env=env, capture_output=True, text=True, timeout=timeout)
    return r


def test_forced_pair_rounds_on_small_repeat_heavy_texts():
    """every round from the third on that may be a pair round is one (any list size, a doubling round in between)"""
    r = _run_child({"JPK_PAIR_SHIFT": "31", "JPK_PAIR_MIN": "2", "JPK_PAIR_GAP": "2", "JPK_PAIR_RATIO": "0", "JPK_PAIR_KEEP": "100"})
    assert "PAIR_OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


def test_forced_pair_rounds_with_plain_byte_keys():
    """7-byte keys (what any block above 128 byte values gets) leave more to the rounds"""
    r = _run_child({"JPK_PAIR_SHIFT": "31", "JPK_PAIR_MIN": "2", "JPK_PAIR_GAP": "3", "JPK_PAIR_RATIO": "0", "JPK_PAIR_KEEP": "100", "JPK_KEY_BITS": "8"}, iters=120, seed=12)
    assert "PAIR_OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


def test_plain_prefix_doubling_remains_a_working_comparator():
    r = _run_child({"JPK_PAIR_SHIFT": "-1"}, iters=60, seed=13)
    assert "PAIR_OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


@pytest.fixture(scope="module")
def gpu():
    import torch
    import jampack_amd as jam
    ctx = jam.Context(0, torch.cuda.current_stream().cuda_stream)
    yield torch, jam, ctx
    ctx.close()


def _fwd(torch, jam, ctx, t):
    dev = torch.device("cuda", 0)
    n = len(t)
    d_in = torch.from_numpy(np.ascontiguousarray(t)).to(dev)
    d_out = torch.empty(n + jam.TRAILER, dtype=torch.uint8, device=dev)
    ctx.bwt_forward(d_in, n, d_out, n + jam.TRAILER)
    torch.cuda.synchronize()
    return d_out.cpu().numpy(), ctx.stats()


@pytest.mark.parametrize("kind,n", [("repeat", 8 << 20), ("repeat", (16 << 20) + 12345), ("repeat4k", 4 << 20), ("silesia", 12 << 20)])
def test_default_knobs_resolve_repeated_segments_in_a_few_rounds(gpu, ref, kind, n):
    """a 1 MiB (4 KiB) text segment repeated: 8-16 copies of every suffix tie until the distance passes the period -- 17+ doubling rounds
    before round 5; the image is the reference's and a pair round was used"""
    torch, jam, ctx = gpu
    t = jam.corpus.make(kind, n, 3)
    got, s = _fwd(torch, jam, ctx, t)
    assert np.array_equal(got, ref.bwt_forward(t))
    if kind != "silesia":
        assert s.sa_pair_rounds != 0 and s.sa_rounds <= 9, (s.sa_rounds, bin(s.sa_pair_rounds), list(s.sa_round_active[: s.sa_rounds]))


def test_a_block_that_holds_everything_twice_takes_the_pair_round_early(gpu, ref):
    """two copies of one text (a tar with the same file twice): every suffix of the first copy ties with its twin for the length of the
    copy, round 1 resolves next to nothing -- 99 % of its list is still there -- and round 2 is a pair round already (JPK_PAIR_EARLY):
    8.6 -> 6.9 ms for 2 x 16 MiB, 40.6 with plain doubling; the reference's bytes"""
    torch, jam, ctx = gpu
    for n_half, kind in ((3 << 20, "text_survey"), (1_500_001, "text_wide")):
        half = jam.corpus.make(kind, n_half, 4)
        t = np.concatenate([half, half, half[: n_half // 3]])
        got, s = _fwd(torch, jam, ctx, t)
        assert np.array_equal(got, ref.bwt_forward(t))
        assert (s.sa_pair_rounds >> 2) & 1 and s.sa_rounds <= 7, (s.sa_rounds, bin(s.sa_pair_rounds), list(s.sa_round_active[: s.sa_rounds]))


def test_group_sort_of_blocks_that_repeat(gpu, oracle):
    """the group sort (several small blocks as one text): stretches stop at the end of their own block, and a copy that ends its block
    is decided by the empty suffix there"""
    torch, jam, ctx = gpu
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(3)
    seg = rng.integers(0, 4, 30_000).astype(np.uint8)
    blocks = [np.concatenate((seg, seg, seg[:20_000])),                       # ends inside a copy
              np.concatenate((rng.integers(0, 4, 1000).astype(np.uint8), seg, rng.integers(0, 4, 1000).astype(np.uint8), seg)),
              np.tile(seg[:777], 200),
              jam.corpus.make("repeat4k", 300_000, 4),
              np.concatenate((seg, seg))]                                      # the same bytes as the head of block 0: ties never cross blocks
    d_in = [torch.from_numpy(np.ascontiguousarray(b)).to(dev) for b in blocks]
    caps = [jam.ans_capacity(len(b) + jam.TRAILER) for b in blocks]
    d_out = [torch.empty(c, dtype=torch.uint8, device=dev) for c in caps]
    n, st = ctx.blocks_compress(d_in, [len(b) for b in blocks], d_out, caps, 2)
    assert st == [0] * len(blocks)
    for i, b in enumerate(blocks):
        want = oracle.ans_encode(oracle.bwt_forward(b))
        assert n[i] == len(want) and np.array_equal(d_out[i][: n[i]].cpu().numpy(), want), i
