"""Parity of the HIP path (through the C ABI) against the oracle and the committed golden vectors.  -m gpu

Bit-exact is the bar for every byte: BWT image + trailer, entropy stream, decoded block.
"""
import numpy as np
import pytest

from golden_util import case_input, cases, sha, small

pytestmark = pytest.mark.gpu

KINDS = ["text", "text_survey", "random", "dna", "two", "zero", "geometric", "samples16", "runs", "repeat4k", "silesia"]
SIZES = [0, 1, 119, 120, 121, 240, 1207, 4097, 70_000, 300_000]


def _first_diff(a, b):
    n = min(len(a), len(b))
    d = np.nonzero(a[:n] != b[:n])[0]
    return f"len {len(a)} vs {len(b)}, first diff at {d[:4].tolist()} of {d.size}"


@pytest.fixture(scope="module")
def jam():
    import jampack_amd
    return jampack_amd


@pytest.mark.parametrize("kind", KINDS)
def test_forward_bwt_equals_oracle(jam, oracle, kind):
    for n in SIZES:
        t = jam.corpus.make(kind, n, 21)
        out = np.full(n + 480, 0x11, dtype=np.uint8)
        got = jam.Bwt().ForwardBwt(t, out=out)
        exp = oracle.bwt_forward(t, prefill=0x11)
        assert np.array_equal(got, exp), f"{kind} n={n}: {_first_diff(got, exp)}"


def _run_cases():
    """inputs built around runs of >= 7 equal bytes (the suffix sort's run members, bwt_fwd.hip RUNF): runs that end in a smaller
    byte, in a larger byte and at the end of the text, of every length around 7 and around the doubling distances, beside each other"""
    rng = np.random.default_rng(4)
    out = {}
    out["one_run_all"] = np.full(2400, 7, np.uint8)
    out["run_then_smaller"] = np.concatenate([np.full(1000, 9, np.uint8), np.full(200, 3, np.uint8)])
    out["run_then_larger"] = np.concatenate([np.full(1000, 9, np.uint8), np.full(200, 200, np.uint8)])
    lens = [1, 2, 6, 7, 8, 13, 14, 15, 27, 28, 29, 55, 56, 57, 100, 1000, 5000]
    parts = []
    for k, L in enumerate(lens * 3):
        parts.append(np.full(L, 50, np.uint8))
        parts.append(np.array([[10], [90], [10, 90], [90, 10]][k % 4], np.uint8))
    out["mixed_exits"] = np.concatenate(parts)
    out["ends_in_run"] = np.concatenate([out["mixed_exits"], np.full(777, 50, np.uint8)])
    out["periodic_runs"] = np.tile(np.concatenate([np.zeros(300, np.uint8), np.ones(1, np.uint8)]), 40)
    out["two_symbol_runs"] = np.repeat(rng.integers(0, 2, 400).astype(np.uint8) * 255, rng.integers(1, 40, 400))
    out["equal_runs_different_tails"] = np.concatenate([np.concatenate([np.full(64, 33, np.uint8), rng.integers(34, 40, 5).astype(np.uint8)]) for _ in range(60)])
    out["zero_120k"] = np.zeros(120_000, np.uint8)
    out["ff_then_text"] = None
    return out


@pytest.mark.parametrize("name", sorted(_run_cases()))
def test_forward_bwt_of_run_inputs_equals_oracle(jam, oracle, name):
    t = _run_cases()[name]
    if t is None:
        t = np.concatenate([np.full(30_000, 255, np.uint8), jam.corpus.make("text", 50_000, 3), np.zeros(20_000, np.uint8)])
    for cut in (0, 1, 61):                              # the raw tail (len % 120) moves with the cut
        u = t[: len(t) - cut]
        got = jam.Bwt().ForwardBwt(u)
        exp = oracle.bwt_forward(u)
        assert np.array_equal(got, exp), f"{name} n={len(u)}: {_first_diff(got, exp)}"


@pytest.mark.parametrize("kind", KINDS)
def test_inverse_bwt_equals_input(jam, oracle, kind):
    for n in SIZES:
        t = jam.corpus.make(kind, n, 22)
        b = oracle.bwt_forward(t)
        got = jam.Bwt().InverseBwt(b, threads=8, gpu=True)
        assert np.array_equal(got, t), f"{kind} n={n}: {_first_diff(got, t)}"


@pytest.mark.parametrize("kind", KINDS)
def test_rank_coding_equals_oracle(jam, oracle, kind):
    for n in [0, 1, 2, 100, 4096, 4097, 70_000, 1 << 20]:
        t = jam.corpus.make(kind, n, 23)
        r, f = jam.Postcoder().Encode(t)
        er, ef = oracle.rank_encode(t)
        assert np.array_equal(f, ef), f"{kind} n={n}: freq differs"
        assert np.array_equal(r, er), f"{kind} n={n}: {_first_diff(r, er)}"
        back = jam.Postcoder().Decode(er, ef)
        assert np.array_equal(back, t), f"{kind} n={n} decode: {_first_diff(back, t)}"


@pytest.mark.parametrize("distinct", [1, 2, 63, 64, 65, 127, 128, 129, 191, 192, 193, 255, 256])
def test_rank_coding_at_the_list_register_boundaries(jam, oracle, distinct):
    """k_enc_mtf keeps the recency list in four 64-lane registers: alphabets that end exactly at, one short of and one past a
    register boundary, walked (a) round robin (every rank = distinct - 1: the deepest position, first occurrences enter at the
    first free one), (b) at random with runs, (c) round robin in descending symbol order after a prefix that has seen only half
    of the alphabet (tiles that start with a partly filled list); lengths that cross several 4 KiB tiles, ragged at the end."""
    rng = np.random.default_rng(1000 + distinct)
    syms = rng.permutation(256)[:distinct].astype(np.uint8)
    n = 3 * 4096 + 17
    cases = {
        "round_robin": syms[np.arange(n) % distinct],
        "random_runs": np.repeat(syms[rng.integers(0, distinct, n)], rng.integers(1, 4, n))[:n],
        "half_then_all": np.concatenate([syms[: max(1, distinct // 2)][np.arange(5000) % max(1, distinct // 2)],
                                         syms[::-1][np.arange(n - 5000) % distinct]]),
    }
    for name, t in cases.items():
        t = np.ascontiguousarray(t, dtype=np.uint8)
        r, f = jam.Postcoder().Encode(t)
        er, ef = oracle.rank_encode(t)
        assert np.array_equal(f, ef), f"{name}: freq differs"
        assert np.array_equal(r, er), f"{name}: {_first_diff(r, er)}"
        assert np.array_equal(jam.Postcoder().Decode(er, ef), t), name


@pytest.mark.parametrize("kind", KINDS)
def test_ans_encode_equals_oracle(jam, oracle, kind):
    for n in [0, 1, 100, 4097, 70_000, 1 << 20, (1 << 20) + 480, 2_500_000]:
        x = oracle.bwt_forward(jam.corpus.make(kind, n, 24)) if n else np.zeros(0, dtype=np.uint8)
        got = jam.Ans().Encode(x)
        exp = oracle.ans_encode(x)
        assert np.array_equal(got, exp), f"{kind} n={n}: {_first_diff(got, exp)}"


@pytest.mark.parametrize("spacing", [1500, 3000, 6000, 20000, 0])
def test_ans_encode_with_quiet_stretches_between_rare_classes(jam, oracle, spacing):
    """The adaptive exponent entries are solved in 4096-symbol segments after a 1280-symbol warm-up (k_adapt_a / ext / tab / b / c).
    A stream whose symbols stay in one low class for thousands of symbols and then shows ONE symbol of a high class exercises
    the cases apart: spacing 0 = never (every segment of the upper entries is the identity), 20000 = quiet warm-up and quiet
    segment mostly, 6000 / 3000 = a quiet warm-up and one visitor inside the segment (the two ends of the interval meet), 1500 = a
    visitor in most warm-ups.  Two alternating bytes give rank 1 throughout; the visitors rotate through 200 other bytes, so their
    ranks -- and classes -- are large."""
    n = (1 << 20) + 4321
    t = np.where(np.arange(n) % 2 == 0, 7, 9).astype(np.uint8)
    if spacing:
        pos = np.arange(spacing, n, spacing)
        t[pos] = (20 + (np.arange(len(pos)) * 37) % 200).astype(np.uint8)
    got = jam.Ans().Encode(t)
    exp = oracle.ans_encode(t)
    assert np.array_equal(got, exp), _first_diff(got, exp)
    assert np.array_equal(jam.Ans().Decode(exp, len(t), threads=4), t)


@pytest.mark.parametrize("kind", KINDS)
def test_ans_decode_equals_oracle(jam, oracle, kind):
    for n in [0, 1, 100, 4097, 70_000, (1 << 20) + 480, 2_500_000]:
        x = oracle.bwt_forward(jam.corpus.make(kind, n, 25)) if n else np.zeros(0, dtype=np.uint8)
        enc = oracle.ans_encode(x)
        got = jam.Ans().Decode(enc, len(x), threads=4)
        assert np.array_equal(got, x), f"{kind} n={n}: {_first_diff(got, x)}"


@pytest.mark.parametrize("kind", ["text", "geometric", "silesia", "zero"])
def test_block_pipeline_equals_oracle(jam, oracle, kind):
    for n in [0, 50, 5000, 1_300_000]:
        t = jam.corpus.make(kind, n, 26)
        comp = jam.block_compress(t)
        if n >= 120:      # below 120 bytes the reference codes an uninitialised trailer (bwt.cpp:35)
            assert np.array_equal(comp, oracle.compress_block(t)), f"{kind} n={n}"
        back = jam.block_decompress(comp, n)
        assert np.array_equal(back, t), f"{kind} n={n}: {_first_diff(back, t)}"


@pytest.mark.parametrize("case", cases(), ids=lambda c: c["name"])
def test_golden_vectors(jam, case):
    t = case_input(case)
    out = np.full(len(t) + 480, 0xAB, dtype=np.uint8)
    bwt = jam.Bwt().ForwardBwt(t, out=out)
    assert sha(bwt) == case["bwt_sha256"], "ForwardBwt differs from the reference"
    ans = jam.Ans().Encode(bwt)
    assert len(ans) == case["ans_len"] and sha(ans) == case["ans_sha256"], "Ans::Encode differs from the reference"
    if case["raw"]:
        assert np.array_equal(ans, small()[case["name"] + ".ans"])
    dec = jam.Ans().Decode(ans, len(bwt))
    assert sha(dec) == case["bwt_sha256"], "Ans::Decode differs"
    back = jam.Bwt().InverseBwt(dec)
    assert np.array_equal(back, t), "InverseBwt differs"
    r, f = jam.Postcoder().Encode(bwt[: 1 << 20])
    assert sha(r) == case["rank0_sha256"] and sha(f.astype("<i4")) == case["freq0_sha256"]


def test_error_codes(jam):
    from jampack_amd import JampackError
    with pytest.raises(JampackError) as e:
        jam.Ans().Encode(np.zeros(100000, dtype=np.uint8) + np.arange(100000, dtype=np.uint8), cap=64)
    assert e.value.status == -2            # JPK_E_CAPACITY instead of the reference's buffer overflow
    with pytest.raises(JampackError) as e:
        jam.Ans().Decode(np.full(300, 0x80, dtype=np.uint8), 1000)
    assert e.value.status == -3
    bad = np.zeros(1200 + 480, dtype=np.uint8)
    bad[1200:1204] = np.frombuffer(np.int32(5000).tobytes(), dtype=np.uint8)     # primary index out of range
    with pytest.raises(JampackError) as e:
        jam.Bwt().InverseBwt(bad)
    assert e.value.status == -3
