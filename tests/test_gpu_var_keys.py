"""Variable-length keys of the suffix sort's round 0 (bwt_fwd.hip k_key_plan / pack_tile_var; divsufsort.cpp:1427-1520 does not care how
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
        assert s.sa_key_depth == depth, (sigma, s.sa_key_depth)


def test_text_kinds_and_the_round_structure(gpu, ref):
    """the bench's two texts at 8 MiB: the reference's bytes, deeper keys than the fixed code's, fewer suffixes left for round 1"""
    torch, jam, ctx = gpu
    for kind, fixed_depth in (("text_survey", 11), ("text_wide", 7)):
        t = jam.corpus.make(kind, 8 << 20, 8)
        got, s = _fwd(torch, jam, ctx, t)
        assert np.array_equal(got, ref.bwt_forward(t, prefill=0x11))
        assert s.sa_key_depth > fixed_depth, (kind, s.sa_key_depth)


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
from test_gpu_var_keys import _skewed
o = Oracle()
ok = True
for sigma, n in ((129, 70_001), (205, 200_000), (256, 4097), (28, 120_000), (6, 5000)):
    t = _skewed(sigma, n, 3 * sigma)
    ok = ok and np.array_equal(jam.Bwt().ForwardBwt(t), o.bwt_forward(t))
for kind, n in (("text_wide", 300_000), ("silesia", 400_000), ("runs", 200_000), ("repeat4k", 100_000)):
    t = jam.corpus.make(kind, n, 3)
    ok = ok and np.array_equal(jam.Bwt().ForwardBwt(t), o.bwt_forward(t))
print("CMP_OK" if ok else "CMP_BAD")
"""


@pytest.mark.parametrize("env", [{"JPK_VARKEYS": "0"}, {"JPK_KEY_BITS": "8"}, {"JPK_ONESWEEP": "0"}])
def test_fixed_width_forms_remain_working_comparators(env):
    """JPK_VARKEYS=0: the alphabet-packed fixed-width keys of round 4; JPK_KEY_BITS=8: plain bytes; JPK_ONESWEEP=0: the two-pass radix
    (which has no room for the depth tag: fixed-width keys)"""
    r = subprocess.run([sys.executable, "-c", _CHILD % {"root": ROOT}], env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
    assert "CMP_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
