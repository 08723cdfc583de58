"""Randomised differential parity: structured random blocks (mixtures of runs, repeats, noise and small alphabets at random
sizes) through the fused GPU entry points against the oracle, byte for byte, plus the round trip.  -m gpu
Complements the fixed corpora: it reaches the rare paths of the rewritten entropy kernels (classes 6/7, model rebuilds at
odd positions, rank-decode row top-ups, bucket ends inside a run, chunk boundaries at arbitrary places)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def make_block(rng, n):
    out = np.empty(n, dtype=np.uint8)
    pos = 0
    hist = []
    while pos < n:
        kind = rng.integers(0, 7)
        ln = int(min(n - pos, rng.integers(1, 1 + max(1, n // 3))))
        if kind == 0:                                        # run of one byte
            seg = np.full(ln, rng.integers(0, 256), dtype=np.uint8)
        elif kind == 1:                                      # noise over a random alphabet size
            a = int(rng.integers(1, 257))
            seg = rng.integers(0, a, ln, dtype=np.int64).astype(np.uint8)
        elif kind == 2 and hist:                             # copy of an earlier piece (long repeats)
            s, l0 = hist[rng.integers(len(hist))]
            reps = -(-ln // l0)
            seg = np.tile(out[s:s + l0], reps)[:ln]
        elif kind == 3:                                      # short period
            p = rng.integers(0, 256, int(rng.integers(1, 40)), dtype=np.int64).astype(np.uint8)
            seg = np.tile(p, -(-ln // len(p)))[:ln]
        elif kind == 4:                                      # geometric-ish small values
            seg = np.minimum(rng.geometric(0.3, ln) - 1, 255).astype(np.uint8)
        elif kind == 5:                                      # ramp / sawtooth
            seg = (np.arange(ln) * int(rng.integers(1, 9)) + int(rng.integers(0, 256))).astype(np.uint8)
        else:                                                # two-symbol noise
            a, b = rng.integers(0, 256, 2)
            seg = np.where(rng.random(ln) < rng.random(), a, b).astype(np.uint8)
        out[pos:pos + ln] = seg
        hist.append((pos, ln))
        pos += ln
    return out


@pytest.fixture(scope="module")
def jam():
    import jampack_amd
    return jampack_amd


@pytest.mark.parametrize("seed", range(40))
def test_random_blocks_match_the_oracle(jam, oracle, seed):
    rng = np.random.default_rng(1000 + seed)
    sizes = [int(rng.integers(0, 400)), int(rng.integers(120, 70_000)), int(rng.integers(70_000, 600_000))]
    if seed % 4 == 0:
        sizes.append(int(rng.integers(1 << 20, (1 << 20) + 300_000)))       # crosses a chunk boundary
    for n in sizes:
        t = make_block(rng, n)
        comp = jam.block_compress(t)
        if n >= 120:                                       # below 120 bytes the trailer bytes are unspecified (bwt.cpp:35)
            exp = oracle.compress_block(t)
            assert np.array_equal(comp, exp), (seed, n, len(comp), len(exp))
        back = jam.block_decompress(comp, n)
        assert np.array_equal(back, t), (seed, n)
