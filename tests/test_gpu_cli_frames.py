"""Frames written by the stock reference CLI (Jampack::Comp with all six stages + CompWriteBlock) decode end to end:
entropy decode + inverse BWT on the GPU, LZ77 / LPX / filter decoders on the host.  -m gpu"""
import json
import os

import numpy as np
import pytest

from golden_util import GOLD

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def jam():
    import jampack_amd
    return jampack_amd


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(GOLD, "golden_cli.npz")), json.load(open(os.path.join(GOLD, "golden_cli_manifest.json")))


def test_golden_frames_of_the_reference_cli(jam, golden):
    z, man = golden
    for c in man["frames"]:
        t = jam.corpus.make(c["kind"], c["n"], c["seed"])
        frame = z[c["name"]]
        assert len(frame) == c["frame_len"]
        back, used = jam.jam_cli_block_read(frame, c["block_size"])
        assert used == len(frame), c["name"]
        assert np.array_equal(back, t), c["name"]
        assert jam.Checksum().IntegrityCheck(back) == c["crc"] == jam.checksum_host(back)


def test_golden_two_block_stream(jam, golden):
    z, man = golden
    s = man["stream"]
    t = jam.corpus.make(s["kind"], s["n"], s["seed"])
    assert np.array_equal(jam.jam_cli_decompress(z[s["name"]]), t)


@pytest.mark.parametrize("kind,n,mf,fl", [("text", 180_000, 0, 1), ("samples16", 90_000, 0, 1), ("samples16", 40_000, 0, 2),
                                          ("silesia", 220_000, 0, 1), ("runs", 120_000, 1, 1), ("repeat4k", 100_000, 1, 1),
                                          ("text", 1 << 20, 0, 1)])
def test_live_reference_frames(jam, ref, kind, n, mf, fl):
    t = jam.corpus.make(kind, n, 81)
    frame = ref.jam_comp_block(t, 1 << 20, mf, fl)
    back, used = jam.jam_cli_block_read(frame, 1 << 20)
    assert used == len(frame)
    assert np.array_equal(back, t)


def test_cli_frame_corruption_is_detected(jam, golden):
    z, man = golden
    c = man["frames"][0]
    f = z[c["name"]].copy()
    bad = f.copy(); bad[4] ^= 0x10                   # crc
    with pytest.raises(jam.JampackError) as e:
        jam.jam_cli_block_read(bad, c["block_size"])
    assert e.value.status == -3
    bad = f.copy(); bad[len(f) // 2] ^= 0x04          # payload
    with pytest.raises(jam.JampackError):
        jam.jam_cli_block_read(bad, c["block_size"])
    with pytest.raises(jam.JampackError):
        jam.jam_cli_block_read(f[: len(f) - 3], c["block_size"])
