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
