"""Full-size blocks (BASELINE.json configs 2, 3, 5) on the GPU: size-independent properties (round trip through every
stage, trailer / length invariants, checksum of the concatenated stream) and -- when the reference build
oracle/_ref is present -- byte equality with the reference itself on the same block.  -m gpu"""
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


def _roundtrip(torch, jam, ctx, t):
    n = len(t)
    dev = torch.device("cuda", 0)
    d_in = torch.from_numpy(t).to(dev)
    cap = jam.ans_capacity(n + 480)
    d_bwt = torch.empty(n + 480, dtype=torch.uint8, device=dev)
    d_enc = torch.empty(cap, dtype=torch.uint8, device=dev)
    d_dec = torch.empty(n + 480, dtype=torch.uint8, device=dev)
    d_back = torch.empty(n, dtype=torch.uint8, device=dev)
    assert ctx.bwt_forward(d_in, n, d_bwt, n + 480) == n + 480
    # invariants of the image: trailer[k] = ISA[k*step]+1 in [1, nlen], trailer values distinct, tail raw
    nlen = n - n % 120
    trailer = d_bwt[n:].cpu().numpy().view("<i4")
    assert trailer.min() >= 1 and trailer.max() <= nlen and len(set(trailer.tolist())) == 120
    assert torch.equal(d_bwt[nlen:n], d_in[nlen:n])
    # the BWT is a permutation of the sorted part: byte histograms agree
    assert torch.equal(torch.bincount(d_bwt[:nlen].int(), minlength=256), torch.bincount(d_in[:nlen].int(), minlength=256))
    clen = ctx.ans_encode(d_bwt, n + 480, d_enc, cap)
    assert ctx.ans_decode(d_enc, clen, d_dec, n + 480) == n + 480
    assert torch.equal(d_dec, d_bwt)
    assert ctx.bwt_inverse(d_dec, n + 480, d_back, n) == n
    assert torch.equal(d_back, d_in)
    # fused entry points give the same stream
    d_enc2 = torch.empty(cap, dtype=torch.uint8, device=dev)
    assert ctx.block_compress(d_in, n, d_enc2, cap) == clen
    assert torch.equal(d_enc2[:clen], d_enc[:clen])
    assert ctx.block_decompress(d_enc2, clen, d_back, n) == n
    assert torch.equal(d_back, d_in)
    return d_bwt, d_enc[:clen]


def test_enwik8_like_64mib_block(gpu, ref):
    torch, jam, ctx = gpu
    t = jam.corpus.make("text", 100_000_000, 8)[: 64 << 20]
    d_bwt, d_enc = _roundtrip(torch, jam, ctx, t)
    # bit-exact against the reference on the full 64 MiB block
    rb = ref.bwt_forward(t)
    assert hashlib.sha256(rb.tobytes()).hexdigest() == _sha(d_bwt)
    re_ = ref.ans_encode(rb)
    assert len(re_) == d_enc.numel() and hashlib.sha256(re_.tobytes()).hexdigest() == _sha(d_enc)


def test_enwik8_like_tail_block(gpu):
    torch, jam, ctx = gpu
    t = jam.corpus.make("text", 100_000_000, 8)[64 << 20:]
    assert len(t) == 32_891_136
    _roundtrip(torch, jam, ctx, t)


def _same_as_reference(ref, t, d_bwt, d_enc):
    """SHA-256 of the BWT image and of the rANS stream against the reference build on the same block (bwt.cpp:22-65, ans.cpp:113-234)"""
    rb = ref.bwt_forward(t)
    assert hashlib.sha256(rb.tobytes()).hexdigest() == _sha(d_bwt)
    re_ = ref.ans_encode(rb)
    assert len(re_) == d_enc.numel() and hashlib.sha256(re_.tobytes()).hexdigest() == _sha(d_enc)


def test_bench_block_64mib_text_survey(gpu, ref):
    """the bench line's own first block (workload enwik8 = SURVEY 8d's text model, 28 byte values: 11-byte keys)"""
    torch, jam, ctx = gpu
    t, _ = jam.corpus.load_or_make("enwik8", start=0, count=64 << 20)
    assert len(t) == 64 << 20
    d_bwt, d_enc = _roundtrip(torch, jam, ctx, t)
    assert ctx.stats().sa_key_depth >= 11          # 11 bytes with the fixed 5-bit code of its 28 byte values, 12 on average with the variable-length code
    _same_as_reference(ref, t, d_bwt, d_enc)


@pytest.mark.parametrize("kind", ["text_wide", "silesia", "samples16", "text"])
def test_wide_alphabet_64mib_blocks_equal_the_reference(gpu, ref, kind):
    """the path real data takes -- more than 128 byte values: 7-byte keys, five and more rounds -- compared with the reference's BYTES at
    full size: an enwik8-like alphabet (207 values), the mixed block, 16-bit samples; and the phrase-book text (28 letters,
    verbatim phrases: five rounds behind deep order-2 keys)"""
    torch, jam, ctx = gpu
    t = jam.corpus.make(kind, 64 << 20, 5)
    d_bwt, d_enc = _roundtrip(torch, jam, ctx, t)
    s = ctx.stats()
    assert 7 <= s.sa_key_depth <= 20, s.sa_key_depth      # 7 bytes with the fixed 8-bit code; the variable-length codes hold 10 (order 0) to 14+ (order 2) symbols
    _same_as_reference(ref, t, d_bwt, d_enc)


def test_silesia_like_212mb_block(gpu, ref):
    """config 5: one mixed block (text, 16-bit samples, random, DNA, runs, 1 MiB segment repeated) of 211 938 580 B, 256 byte values:
    the reference's bytes (15 s of host time), and the repeated segment no longer costs log2(LCP) doubling rounds (round 4: 21)"""
    torch, jam, ctx = gpu
    t = jam.corpus.make("silesia", 211_938_580, 5)
    d_bwt, d_enc = _roundtrip(torch, jam, ctx, t)
    s = ctx.stats()
    assert s.sa_rounds <= 12 and s.sa_pair_rounds != 0, (s.sa_rounds, bin(s.sa_pair_rounds))
    _same_as_reference(ref, t, d_bwt, d_enc)


def test_grouped_8mib_wide_alphabet_blocks_equal_the_reference(gpu, ref):
    """the reference's default block size through the grouped path of jpk_dev_blocks_compress (one suffix sort per group), wide alphabet"""
    torch, jam, ctx = gpu
    dev = torch.device("cuda", 0)
    data = jam.corpus.make("text_wide", 3 * (8 << 20) + 12345, 21)
    blocks = jam.corpus.split_blocks(data, 8 << 20)
    d_in = [torch.from_numpy(np.ascontiguousarray(b)).to(dev) for b in blocks]
    caps = [jam.ans_capacity(len(b) + 480) for b in blocks]
    d_out = [torch.empty(c, dtype=torch.uint8, device=dev) for c in caps]
    n, st = ctx.blocks_compress(d_in, [len(b) for b in blocks], d_out, caps, 2)
    assert st == [0] * len(blocks)
    for i, b in enumerate(blocks):
        want = ref.ans_encode(ref.bwt_forward(b))
        assert n[i] == len(want) and hashlib.sha256(want.tobytes()).hexdigest() == _sha(d_out[i][: n[i]]), i


def test_worst_case_blocks_16mib(gpu, oracle):
    torch, jam, ctx = gpu
    for kind in ("zero", "repeat", "two"):
        t = jam.corpus.make(kind, 16 << 20, 3)
        d_bwt, _ = _roundtrip(torch, jam, ctx, t)


def test_120_chain_comparator_gives_the_same_text(gpu):
    """BASELINE config 3's "120-way parallel LF-map": the reference's own kernel shape (CUDAInverse<<<40,3>>>, bwt.cpp:8-19)
    kept as a measured comparator; same bytes as the list-ranking inverse, on a block small enough for 120 serial chains."""
    torch, jam, ctx = gpu
    n = 2_400_000 + 77
    t = jam.corpus.make("text_survey", n, 4)
    dev = torch.device("cuda", 0)
    d_in = torch.from_numpy(t).to(dev)
    d_bwt = torch.empty(n + 480, dtype=torch.uint8, device=dev)
    ctx.bwt_forward(d_in, n, d_bwt, n + 480)
    a = torch.zeros(n, dtype=torch.uint8, device=dev)
    b = torch.zeros(n, dtype=torch.uint8, device=dev)
    assert ctx.bwt_inverse(d_bwt, n + 480, a, n) == n
    m, ms = ctx.bwt_inverse_chains120(d_bwt, n + 480, b, n)
    assert m == n and ms > 0 and torch.equal(a, b) and torch.equal(a, d_in)
