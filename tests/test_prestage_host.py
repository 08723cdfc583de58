"""Host-side pre-stage decoders (Lz77::Decompress, Lpx::Decode, Filters::Decode, checksum) against the
real reference build (oracle/_ref) and the committed golden streams.  CPU only: these entry points never touch the GPU."""
import json
import os

import numpy as np
import pytest

from golden_util import GOLD as GOLDEN_DIR

KINDS = ["text", "samples16", "repeat4k", "random", "runs", "geometric"]
SIZES = [0, 1, 3, 4, 5, 17, 1000, 65536, 65537, 100_003]


@pytest.fixture(scope="module")
def jam():
    import jampack_amd
    return jampack_amd


@pytest.mark.parametrize("kind", KINDS)
def test_lpx_decode_matches_reference(jam, ref, kind):
    for n in SIZES:
        if n < 4:
            continue                     # the reference's part loop never ends / divides by zero for len < 4 (lpx.cpp:148, 160)
        t = jam.corpus.make(kind, n, 51)
        enc = ref.lpx_encode(t)
        got = jam.Lpx().Decode(enc)
        assert np.array_equal(got, ref.lpx_decode(enc)), (kind, n)
        assert np.array_equal(got, t), (kind, n)


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("filters", [1, 2])
def test_filters_decode_matches_reference(jam, ref, kind, filters):
    for n in SIZES:
        if filters == 2 and n > 20_000:
            continue                     # brute-force mode of the reference encoder is slow
        t = jam.corpus.make(kind, n, 52)
        enc = ref.filters_encode(t, filters)
        got = jam.Filters().Decode(enc, n + 64)
        assert np.array_equal(got, ref.filters_decode(enc, n + 64)), (kind, n)
        assert np.array_equal(got, t), (kind, n)


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("mf", [0, 1, 4])
def test_lz77_decompress_matches_reference(jam, ref, kind, mf):
    for n in [0, 1, 5, 1000, 65537, 120_000]:
        if (mf and n > 70_000) or (mf > 1 and (n < 65_537 or kind not in ("text", "repeat4k", "runs"))):
            continue                     # the reference's deeper match finders are slow on noisy data, and crash on small inputs
        t = jam.corpus.make(kind, n, 53)
        enc = ref.lz77_compress(t, mf)
        got = jam.Lz77().Decompress(enc, n + 64)
        assert np.array_equal(got, ref.lz77_decompress(enc, n + 64)), (kind, n, mf)
        assert np.array_equal(got, t), (kind, n, mf)


def test_lpx_decode_tiny_inputs_pass_through(jam):
    for n in range(0, 4):
        t = jam.corpus.make("text", n, 55)
        assert np.array_equal(jam.Lpx().Decode(t), t)


def test_checksum_host_matches_reference(jam, ref):
    t = jam.corpus.make("text", 100_000, 54)
    for n in list(range(0, 70)) + [4096, 65537, 100_000]:
        assert jam.checksum_host(t[:n]) == ref.checksum(t[:n]), n


def test_prestage_decoders_reject_bad_streams(jam):
    lz = jam.Lz77()
    with pytest.raises(jam.JampackError):                 # match before the start of the output
        lz.Decompress(np.array([0x00, 0x85], dtype=np.uint8), 100)
    with pytest.raises(jam.JampackError):                 # truncated offset
        lz.Decompress(np.array([0x08, 0x01], dtype=np.uint8), 100)
    with pytest.raises(jam.JampackError):                 # output capacity
        lz.Decompress(np.array([0x04, 0x80] + [7] * 50, dtype=np.uint8), 10)
    fl = jam.Filters()
    with pytest.raises(jam.JampackError):                 # unsupported filter type (filters.cpp:455)
        fl.Decode(np.array([3, 1, 0, 0], dtype=np.uint8), 100)
    with pytest.raises(jam.JampackError):                 # channel width > 32
        fl.Decode(np.array([0, 33, 0, 0], dtype=np.uint8), 100)


def test_golden_prestage_streams(jam):
    """streams produced by the reference encoders, committed as data (tests/golden/make_golden_cli.py)"""
    z = np.load(os.path.join(GOLDEN_DIR, "golden_cli.npz"))
    man = json.load(open(os.path.join(GOLDEN_DIR, "golden_cli_manifest.json")))
    for c in man["stages"]:
        t = jam.corpus.make(c["kind"], c["n"], c["seed"])
        enc = z[c["name"]]
        if c["stage"] == "lpx":
            got = jam.Lpx().Decode(enc)
        elif c["stage"] == "filters":
            got = jam.Filters().Decode(enc, c["n"] + 64)
        else:
            got = jam.Lz77().Decompress(enc, c["n"] + 64)
        assert np.array_equal(got, t), c["name"]
