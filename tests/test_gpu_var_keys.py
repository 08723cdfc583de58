"""Variable-length keys of the suffix sort's round 0 (bwt_fwd.hip k_key_plan / k_ctx_plan / k_key_final / k_pack_keys_var; divsufsort.cpp:1427-1520 does not care how
many byte values a block uses -- a fixed-width code does: 7 bytes per key above 128 values).  An order-preserving prefix code built
from the block's (sampled) histogram packs about 56 / H0 symbols into a key and every group of tied suffixes carries its own depth.
The forward BWT must stay the reference's (bwt.cpp:22-65): skewed alphabets of 129..256 values, codes that straddle the 56-bit boundary
at every phase, runs of frequent and of rare bytes, the end of the text inside a key, and the fixed-width forms as comparators.  -m gpu"""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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
    d_out = torch.full((n + jam.TRAILER,), 0x11, dtype=torch.uint8, device=dev)
    ctx.bwt_forward(d_in, n, d_out, n + jam.TRAILER)
    torch.cuda.synchronize()
    return d_out.cpu().numpy(), ctx.stats()


def _skewed(sigma, n, seed, s=1.2):
    """n bytes over `sigma` byte values (always 0 and 255 among them) with Zipf(s) frequencies in a shuffled order: code lengths from
    1-2 bits to 14+, so that the 56-bit boundary falls inside codes of every length; planted repeats, runs of the most and of the
    least frequent byte, a tail that ends inside a run"""
    rng = np.random.default_rng(seed)
    inner = rng.choice(np.arange(1, 255), sigma - 2, replace=False)
    sym = np.sort(np.concatenate(([0, 255], inner))).astype(np.uint8)
    p = np.arange(1, sigma + 1, dtype=np.float64) ** -s
    p /= p.sum()
    perm = rng.permutation(sigma)                       # which byte value is frequent: not the small ones
    t = sym[perm[rng.choice(sigma, n, p=p)]]
    frequent, rare = sym[perm[0]], sym[perm[-1]]
    pos = 100
    for L in (3, 8, 9, 10, 11, 12, 13, 20, 27, 28, 29, 40, 55, 56, 57, 58, 100, 300, 1000):
        if pos + 3 * L + 50 >= n:
            break
        t[pos: pos + L] = frequent
        pos += L + 7
        t[pos: pos + L] = rare
        pos += L + 5
        seg = t[pos: pos + L].copy()
        t[pos + L + 3: pos + 2 * L + 3] = seg           # a repeat at distance L + 3
        pos += 2 * L + 11
    if n > 600:
        t[n - 90:] = frequent                           # the text ends inside a run (keys that run off the end)
        t[n - 400: n - 300] = np.tile(t[n - 400: n - 390], 10)
    return np.ascontiguousarray(t)


@pytest.mark.parametrize("sigma", [129, 160, 205, 256])
def test_skewed_wide_alphabets_equal_the_oracle(gpu, oracle, sigma):
    torch, jam, ctx = gpu
    for n, seed in ((121, 1), (1000, 2), (4097, 3), (70_001, 4), (300_000, 5)):
        t = _skewed(sigma, n, 10 * sigma + seed)
        got, s = _fwd(torch, jam, ctx, t)
        assert np.array_equal(got, oracle.bwt_forward(t, prefill=0x11)), (sigma, n)
        if n >= 70_001:
            assert s.sa_key_depth >= 8, (sigma, n, s.sa_key_depth)          # the fixed 8-bit code holds 7


@pytest.mark.parametrize("sigma,s", [(3, 2.0), (6, 1.5), (20, 1.0), (28, 0.8), (64, 1.3), (100, 2.5)])
def test_skewed_small_alphabets_equal_the_oracle(gpu, oracle, sigma, s):
    torch, jam, ctx = gpu
    for n, seed in ((150, 1), (5000, 2), (120_000, 3)):
        t = _skewed(sigma, n, 7 * sigma + seed, s)
        got, _ = _fwd(torch, jam, ctx, t)
        assert np.array_equal(got, oracle.bwt_forward(t, prefill=0x11)), (sigma, n)


def test_flat_alphabets_keep_the_fixed_width_code(gpu, oracle):
    """a balanced code of a flat histogram is the fixed-width code (and a slightly skewed one can be longer): the plan keeps the fixed
    form unless the variable one holds at least 3/4 of a symbol more per key"""
    torch, jam, ctx = gpu
    rng = np.random.default_rng(1)
    for sigma, depth in ((4, 28), (16, 14), (256, 7)):
        t = rng.integers(0, sigma, 200_000).astype(np.uint8)
        got, s = _fwd(torch, jam, ctx, t)
        assert np.array_equal(got, oracle.bwt_forward(t, prefill=0x11))
        assert s.sa_key_depth == depth and s.sa_key_order == -1, (sigma, s.sa_key_depth, s.sa_key_order)


def test_text_kinds_and_the_round_structure(gpu, ref):
    """the bench's two texts at 8 MiB: the reference's bytes, deeper keys than the fixed code's, fewer suffixes left for round 1"""
    torch, jam, ctx = gpu
    for kind, fixed_depth in (("text_survey", 11), ("text_wide", 7)):
        t = jam.corpus.make(kind, 8 << 20, 8)
        got, s = _fwd(torch, jam, ctx, t)
        assert np.array_equal(got, ref.bwt_forward(t, prefill=0x11))
        assert s.sa_key_depth > fixed_depth, (kind, s.sa_key_depth)


def _markov(n, seed, sigma=60, fan=3, keep=0.93, lo=33):
    """a text with strong order-1 structure: every symbol has `fan` likely successors (the order-1 code's lengths then run from 1-2 bits
    for those to 10+ for the pairs the sample never saw)"""
    rng = np.random.default_rng(seed)
    succ = rng.integers(0, sigma, (sigma, fan))
    r, pick, rnd = rng.random(n), rng.integers(0, fan, n), rng.integers(0, sigma, n)
    out = np.empty(n, dtype=np.uint8)
    s = 0
    for i in range(n):
        s = succ[s, pick[i]] if r[i] < keep else rnd[i]
        out[i] = lo + s
    return out


def test_order1_code_texts_with_context_structure(gpu, oracle):
    """vmode 2 (k_ctx_plan / k_pack_keys_var<true>): every symbol behind a key's first is coded in the context of the byte in front of
    it.  Markov texts (likely successors in 1-2 bits: keys of 25+ symbols, depth tags near the clamp), pairs that occur once in the whole
    text (never in the sample: the floor weight's long codes), runs of a byte that is rare behind itself, a text that ends inside a
    likely chain, 16-bit samples (flat order-0 histogram, all the structure in the pairs)"""
    torch, jam, ctx = gpu
    for n, seed, sigma, fan in ((200, 1, 12, 2), (5000, 2, 60, 3), (70_001, 3, 60, 3), (300_000, 4, 200, 2), (300_000, 5, 8, 1)):
        t = _markov(n, seed, sigma, fan)
        if n > 3000:
            t[1000:1003] = (250, 251, 250)                               # bytes and pairs that occur exactly once
            t[2000:2300] = t[1500]                                       # a run of a byte that is rare behind itself
            t[n - 120:] = t[100: 220]                                    # the text ends inside a copy
        got, s = _fwd(torch, jam, ctx, np.ascontiguousarray(t))
        assert np.array_equal(got, oracle.bwt_forward(np.ascontiguousarray(t), prefill=0x11)), (n, sigma, fan)
        if n >= 70_001:
            assert s.sa_key_depth >= 12, (n, sigma, fan, s.sa_key_depth)
    rng = np.random.default_rng(9)
    smp = (np.cumsum(rng.integers(-300, 301, 150_000)) & 0xFFFF).astype("<u2").view(np.uint8)     # a random walk in 16-bit samples
    got, s = _fwd(torch, jam, ctx, np.ascontiguousarray(smp))
    assert np.array_equal(got, oracle.bwt_forward(np.ascontiguousarray(smp), prefill=0x11))


def _markov2(n, seed, sigma, fan=2, keep=0.95, lo=1):
    """order-2 structure: the likely successors depend on the two bytes in front; with sigma = 200 there are far more than 1024 pair
    contexts, so most symbols fall back to the order-1 rows and the chosen pairs change from key to key"""
    rng = np.random.default_rng(seed)
    succ = rng.integers(0, sigma, (sigma * sigma, fan))
    r, pick, rnd = rng.random(n), rng.integers(0, fan, n), rng.integers(0, sigma, n)
    out = np.empty(n, dtype=np.uint8)
    a = b = 0
    for i in range(n):
        s = succ[a * sigma + b, pick[i]] if r[i] < keep else rnd[i]
        out[i] = lo + s
        a, b = b, s
    return out


def test_order2_code_pairs_of_bytes_as_contexts(gpu, oracle):
    """vmode 3 (k_ctx_select / k_triple_counts / k_pack_keys_o2): symbols from a key's third on are coded behind the TWO bytes in front of
    them when that pair is among the 1024 most frequent of the sample, behind the one byte otherwise.  Few contexts (all chosen), many
    (most symbols in order-1 rows, chosen and other pairs mixed inside a key), near-deterministic chains (1-bit codes: 50+ symbols per key,
    the depth tag at its clamp), runs, unique bytes, texts that end inside a chain, tiny texts without a usable sample"""
    torch, jam, ctx = gpu
    for n, seed, sigma, fan, keep in ((150, 1, 6, 2, 0.9), (3000, 2, 20, 2, 0.95), (70_001, 3, 30, 2, 0.95), (300_000, 4, 200, 2, 0.9),
                                     (200_000, 5, 12, 1, 0.97), (120_000, 6, 90, 3, 0.8)):
        t = _markov2(n, seed, sigma, fan, keep)
        if n > 3000:
            t[1000:1003] = (250, 251, 250)
            t[2000:2400] = t[1500]
            t[5000:5600] = np.tile(t[4000:4003], 200)                    # a period-3 stretch: the same three pair contexts over and over
            t[n - 150:] = t[300: 450]
        t = np.ascontiguousarray(t)
        got, s = _fwd(torch, jam, ctx, t)
        assert np.array_equal(got, oracle.bwt_forward(t, prefill=0x11)), (n, sigma, fan)
        if n >= 70_001 and sigma <= 30:                                  # (90 or 200 byte values: 8 100 or 40 000 pair contexts for 1024 rows)
            assert s.sa_key_depth >= 12 and s.sa_key_order == 2, (n, sigma, fan, s.sa_key_depth, s.sa_key_order)


def test_context_codes_buy_depth_on_the_bench_texts(gpu, ref):
    """the two bench texts at 8 MiB: deeper keys than the order-0 code's (12 and 10 symbols; the order-1 code: 13 and 12), the reference's bytes"""
    torch, jam, ctx = gpu
    for kind, order0_depth in (("text_survey", 13), ("text_wide", 12)):
        t = jam.corpus.make(kind, 8 << 20, 9)
        got, s = _fwd(torch, jam, ctx, t)
        assert np.array_equal(got, ref.bwt_forward(t, prefill=0x11))
        assert s.sa_key_depth > order0_depth and s.sa_key_order == 2, (kind, s.sa_key_depth, s.sa_key_order)


def test_blocks_above_64_mib_clamp_the_depth_tag(gpu, ref):
    """above 2^26 sorted bytes the depth rides in five (four) spare bits of the suffix number: depths are clamped at 31 (15), which is
    still a number of symbols the group shares -- DNA (2-bit codes, 28 symbols per key) makes the clamp bite"""
    torch, jam, ctx = gpu
    rng = np.random.default_rng(7)
    n = (64 << 20) + 4096 + 77
    p = np.array([0.55, 0.25, 0.15, 0.05])
    t = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.choice(4, n, p=p)]
    t[1000:1400] = ord("A")
    t[5_000_000: 5_000_000 + 70_000] = t[100: 100 + 70_000]          # a long repeat
    got, s = _fwd(torch, jam, ctx, np.ascontiguousarray(t))
    assert np.array_equal(got, ref.bwt_forward(np.ascontiguousarray(t), prefill=0x11))


_CHILD = r"""
import sys, numpy as np
sys.path.insert(0, %(root)r)
sys.path.insert(0, %(root)r + "/tests")
import jampack_amd as jam
from oracle.pyoracle import Oracle
from test_gpu_var_keys import _skewed, _markov, _markov2
o = Oracle()
ok = True
for sigma, n in ((129, 70_001), (205, 200_000), (256, 4097), (28, 120_000), (6, 5000)):
    t = _skewed(sigma, n, 3 * sigma)
    ok = ok and np.array_equal(jam.Bwt().ForwardBwt(t), o.bwt_forward(t))
for t in (_markov(120_000, 5), _markov2(120_000, 6, 40)):
    ok = ok and np.array_equal(jam.Bwt().ForwardBwt(t), o.bwt_forward(t))
for kind, n in (("text_wide", 300_000), ("silesia", 400_000), ("runs", 200_000), ("repeat4k", 100_000)):
    t = jam.corpus.make(kind, n, 3)
    ok = ok and np.array_equal(jam.Bwt().ForwardBwt(t), o.bwt_forward(t))
print("CMP_OK" if ok else "CMP_BAD")
"""


@pytest.mark.parametrize("env", [{"JPK_VARKEYS": "0"}, {"JPK_KEY_BITS": "8"}, {"JPK_ONESWEEP": "0"}, {"JPK_KEY_ORDER": "0"}, {"JPK_KEY_ORDER": "1"},
                                 {"JPK_R0_LOOKBACK": "0"}, {"JPK_SA_WAIT_ROUND": "3", "JPK_LG_GRID": "0"}, {"JPK_SA_WAIT_ROUND": "2", "JPK_LG_GRID": "64"}])
def test_fixed_width_forms_remain_working_comparators(env):
    """JPK_VARKEYS=0: the alphabet-packed fixed-width keys of round 4; JPK_KEY_BITS=8: plain bytes; JPK_ONESWEEP=0: the two-pass radix
    (which has no room for the depth tag: fixed-width keys); JPK_KEY_ORDER=0 / 1: nothing above the order-0 / order-1 variable-length code; JPK_R0_LOOKBACK=0: round 0's bookkeeping in two passes
    (k_r0_count + k_r0_scan); JPK_SA_WAIT_ROUND / JPK_LG_GRID: round 4's round control (rounds 1 and 2 enqueued blind, grids on the bound), and a
    tiny large-group grid (every workgroup walks many pieces)"""
    r = subprocess.run([sys.executable, "-c", _CHILD % {"root": ROOT}], env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
    assert "CMP_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
