"""Pins the C restatement against the real reference compiled from /root/reference (oracle/_ref).
Skipped when oracle/_ref/libjamref.so is absent.  CPU only."""
import numpy as np
import pytest

from jampack_amd import corpus

KINDS = ["text", "random", "dna", "two", "zero", "geometric", "samples16", "runs", "repeat4k", "silesia"]
SIZES = [0, 1, 2, 119, 120, 121, 239, 240, 1000, 4097, 70000]


@pytest.mark.parametrize("kind", KINDS)
def test_block_pipeline_equals_reference(oracle, ref, kind):
    for n in SIZES:
        t = corpus.make(kind, n, 21)
        a = oracle.bwt_forward(t, prefill=0x11)
        b = ref.bwt_forward(t, prefill=0x11)
        assert np.array_equal(a, b), (kind, n)
        ea, eb = oracle.ans_encode(a), ref.ans_encode(b)
        assert np.array_equal(ea, eb), (kind, n)
        assert np.array_equal(oracle.ans_decode(eb, len(a)), a)
        assert np.array_equal(ref.ans_decode(ea, len(a), threads=3), a)
        assert np.array_equal(oracle.bwt_inverse(a), t)
        assert np.array_equal(ref.bwt_inverse(a, threads=2), t)


@pytest.mark.parametrize("kind", ["text", "geometric", "random", "runs"])
def test_multi_chunk_equals_reference(oracle, ref, kind):
    t = corpus.make(kind, 2 * (1 << 20) + 12345, 5)
    a, b = oracle.bwt_forward(t), ref.bwt_forward(t)
    assert np.array_equal(a, b)
    ea, eb = oracle.ans_encode(a), ref.ans_encode(b)
    assert np.array_equal(ea, eb)
    assert np.array_equal(oracle.ans_decode(ea, len(a)), a)


def test_suffix_array_equals_divsufsort(oracle, ref):
    for kind in ("text", "two", "zero", "repeat4k", "random"):
        for n in (1, 2, 3, 100, 5000, 200000):
            t = corpus.make(kind, n, 9)
            assert np.array_equal(oracle.suffix_array(t), ref.divsufsort(t)), (kind, n)


def test_stage_functions_equal_reference(oracle, ref):
    rng = np.random.default_rng(1)
    for kind in ("text", "geometric", "random", "zero", "runs"):
        t = corpus.make(kind, 50000, 2)
        ra, fa = oracle.rank_encode(t)
        rb, fb = ref.rank_encode(t)
        assert np.array_equal(ra, rb) and np.array_equal(fa, fb)
        assert np.array_equal(oracle.rank_decode(ra, fa), t)
        assert np.array_equal(ref.rank_decode(ra, fa), t)
        sa, sb = oracle.rle_encode(ra), ref.rle_encode(ra)
        assert np.array_equal(sa, sb)
        assert np.array_equal(oracle.rle_decode(sa, len(ra)), ra)
        assert np.array_equal(ref.rle_decode(sa, len(ra)), ra)
    for v in list(rng.integers(0, 2 ** 31 - 1, 200)) + [0, 126, 127, 16509, 16510, 2113660, 2113661, 270549115, 270549116]:
        assert oracle.leb_encode(int(v)) == ref.leb_encode(int(v))
        assert ref.leb_decode(oracle.leb_encode(int(v)))[0] == int(v)


def test_rans_pairs_roundtrip(oracle):
    t = corpus.make("geometric", 300000, 4)
    r, _ = oracle.rank_encode(t)
    s = oracle.rle_encode(r)
    pairs = oracle.model_pairs(s)
    pay = oracle.rans_encode_pairs(pairs)
    assert np.array_equal(oracle.rans_decode_chunk(pay, len(s)), s)


def test_checksum_matches_reference(oracle, ref):
    from jampack_amd import corpus
    t = corpus.make("text", 200_000, 41)
    for n in list(range(0, 100)) + [4095, 4096, 4097, 65536, 65537, 199_999, 200_000]:
        assert oracle.checksum(t[:n]) == ref.checksum(t[:n]), n
    for kind in ("random", "zero", "two", "geometric"):
        u = corpus.make(kind, 100_003, 42)
        assert oracle.checksum(u) == ref.checksum(u), kind
