"""Blocks above 2^28 bytes on the GPU (format.hpp:20-22 allows blocks up to 1000 MiB).

Up to 2^28 sorted bytes the depth of a slot's variable-length key rides in the spare upper bits of its 32-bit suffix number; above, a
suffix number needs 29 or 30 bits and the depths stay in the slots' own array (bwt_fwd.hip: tag shift 32, r0_short).  Until round 6
such blocks fell back to round 4's fixed-width keys and no test ran one.  Here: a 300 MiB mixed block and the two block sizes next to
the boundary (sorted part 2^28 - 16 and 2^28 + 104 bytes), each compared with the reference build's BYTES -- SHA-256 of the BWT image
(bwt.cpp:22-65) and of the rANS stream (ans.cpp:113-234) -- plus the round trip.  About 25 s of host time per case for the reference's
divsufsort.  -m gpu"""
import hashlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _sha(t):
    return hashlib.sha256(t.cpu().numpy().tobytes()).hexdigest()


@pytest.fixture(scope="module")
def gpu():
    import torch
    import jampack_amd as jam
    assert torch.cuda.is_available()
    ctx = jam.Context(0, torch.cuda.current_stream().cuda_stream)
    yield torch, jam, ctx
    ctx.close()
    jam.lib().jpk_release_idle()


def _compress_and_compare(torch, jam, ctx, ref, t):
    n = len(t)
    dev = torch.device("cuda", 0)
    d_in = torch.from_numpy(t).to(dev)
    cap = jam.ans_capacity(n + 480)
    d_bwt = torch.empty(n + 480, dtype=torch.uint8, device=dev)
    d_enc = torch.empty(cap, dtype=torch.uint8, device=dev)
    assert ctx.bwt_forward(d_in, n, d_bwt, n + 480) == n + 480
    s = ctx.stats()
    clen = ctx.ans_encode(d_bwt, n + 480, d_enc, cap)
    # the reference's bytes
    rb = ref.bwt_forward(t)
    assert hashlib.sha256(rb.tobytes()).hexdigest() == _sha(d_bwt)
    re_ = ref.ans_encode(rb)
    del rb
    assert len(re_) == clen and hashlib.sha256(re_.tobytes()).hexdigest() == _sha(d_enc[:clen])
    # and back
    d_back = torch.empty(n, dtype=torch.uint8, device=dev)
    assert ctx.block_decompress(d_enc, clen, d_back, n) == n
    assert torch.equal(d_back, d_in)
    return s


def test_300_mib_mixed_block_equals_the_reference(gpu, ref):
    """`jampack c -b300`: one block of 314 572 800 bytes of the silesia-like mix (text, 16-bit samples, random bytes, DNA, runs, a 1 MiB
    segment repeated): above 2^28, so the depths of the variable-length keys come from their own array"""
    torch, jam, ctx = gpu
    t = jam.corpus.make("silesia", 300 << 20, 6)
    s = _compress_and_compare(torch, jam, ctx, ref, t)
    assert s.sa_key_order >= 0, "the variable-length keys are in use above 2^28 bytes"
    assert s.sa_rounds <= 14, s.sa_rounds


@pytest.mark.parametrize("nlen", [(1 << 28) - 16, (1 << 28) + 104])
def test_blocks_next_to_the_2_28_boundary_equal_the_reference(gpu, ref, nlen):
    """sorted parts of 2^28 - 16 (the last size whose depths ride in the suffix numbers: four spare bits, depths clamped at 15) and
    2^28 + 104 bytes (the first size whose depths do not); + 7 raw tail bytes (bwt.cpp:32-33).  A text with enwik8's byte alphabet."""
    torch, jam, ctx = gpu
    assert nlen % 120 == 0
    t = jam.corpus.make("text_wide", nlen + 7, 28)
    s = _compress_and_compare(torch, jam, ctx, ref, t)
    assert s.sa_key_order >= 0 and s.sa_key_depth >= 9, (s.sa_key_order, s.sa_key_depth)


def _mixed_block(jam, n, seed):
    """`n` bytes out of DISTINCT 125 MiB parts of the silesia-like mix (a block that repeats one part would be a test of the pair rule)"""
    part = 125 << 20
    out = np.empty(n, dtype=np.uint8)
    for k, lo in enumerate(range(0, n, part)):
        m = min(part, n - lo)
        out[lo:lo + m] = jam.corpus.make("silesia" if k % 3 != 1 else "text_wide", m, seed + k)
    return out


def test_maximum_block_size_round_trip(gpu):
    """`jampack c -b1000`: MAX_BLOCKSIZE (format.hpp:22), 1 048 576 000 bytes in ONE block: suffix numbers of 30 bits, 58 GB of arena, a
    thousand rANS chunks.  The size-independent properties: compress -> decompress is the identity, the stream's header states the sizes,
    and the checksum of the output equals the checksum of the input (checksum.cpp:12-36)."""
    torch, jam, ctx = gpu
    n = jam.api.MAX_BLOCKSIZE
    t = _mixed_block(jam, n, 40)
    dev = torch.device("cuda", 0)
    d_in = torch.from_numpy(t).to(dev)
    del t
    cap = jam.ans_capacity(n + 480)
    d_enc = torch.empty(cap, dtype=torch.uint8, device=dev)
    clen = ctx.block_compress(d_in, n, d_enc, cap)
    s = ctx.stats()
    assert 0 < clen < n // 2, clen
    assert s.sa_key_order >= 0 and s.sa_rounds <= 16, (s.sa_key_order, s.sa_rounds)
    assert jam.ans_decoded_size(d_enc[:clen].cpu().numpy())[0] == n + 480
    d_back = torch.empty(n, dtype=torch.uint8, device=dev)
    assert ctx.block_decompress(d_enc, clen, d_back, n) == n
    assert torch.equal(d_back, d_in)
    assert ctx.checksum(d_back, n) == ctx.checksum(d_in, n)
    # 2^30 bytes would need the rank bits the sort keeps for flags: refused (JPK_FWD_BWT_LIMIT), not sorted wrongly
    del d_back
    big = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
    for call in (ctx.block_compress, ctx.bwt_forward):
        with pytest.raises(jam.JampackError) as e:
            call(big, 1 << 30, d_enc, cap)
        assert e.value.status == -1            # JPK_E_ARG


@pytest.mark.skipif(not __import__("os").environ.get("JPK_TEST_HUGE_REF"), reason="minutes of host time for the reference's divsufsort: JPK_TEST_HUGE_REF=<MiB>")
def test_huge_block_equals_the_reference(gpu, ref):
    """the reference's BYTES for a block far above 2^29 (JPK_TEST_HUGE_REF=600 or 1000): run by hand, recorded in profiles/r06_big_block_max.txt"""
    import os
    torch, jam, ctx = gpu
    n = int(os.environ["JPK_TEST_HUGE_REF"]) << 20
    s = _compress_and_compare(torch, jam, ctx, ref, _mixed_block(jam, n, 50))
    print(f"huge block {n} bytes: rounds {s.sa_rounds}, key order {s.sa_key_order}, arena {s.workspace_bytes >> 20} MiB")
