"""Entropy sub-stage probes (RLE0, model pass) on device buffers vs the oracle.  -m gpu"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import torch
    from jampack_amd import Context
    c = Context(0, torch.cuda.current_stream().cuda_stream)
    yield c
    c.close()


def _dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("kind", ["text", "zero", "geometric", "random", "runs", "two"])
@pytest.mark.parametrize("n", [1, 15, 16, 17, 4096, 4097, 8191, 70_000, 1 << 20])
def test_rle_encode_equals_oracle(ctx, oracle, kind, n):
    import torch
    from jampack_amd import corpus
    r, _ = oracle.rank_encode(corpus.make(kind, n, 31))
    exp = oracle.rle_encode(r)
    d_r = _dev(r)
    d_o = torch.zeros(n + 8, dtype=torch.int16, device="cuda")
    rlen = ctx.rle_encode(d_r, n, d_o)
    got = d_o.cpu().numpy().view(np.uint16)[:rlen]
    assert rlen == len(exp), f"{kind} n={n}: rlen {rlen} vs {len(exp)}"
    bad = np.nonzero(got != exp)[0]
    assert bad.size == 0, f"{kind} n={n}: first mismatch at {bad[:5]}"


@pytest.mark.parametrize("kind", ["text", "geometric", "random", "runs", "zero"])
@pytest.mark.parametrize("n", [1, 9, 10, 100, 5000, 300_000, 1 << 20])
def test_model_pairs_equal_oracle(ctx, oracle, kind, n):
    import torch
    from jampack_amd import corpus
    r, _ = oracle.rank_encode(corpus.make(kind, n, 32))
    s = oracle.rle_encode(r)
    if len(s) == 0:
        pytest.skip("empty")
    exp = oracle.model_pairs(s)
    d_s = _dev(s.view(np.int16))
    d_p = torch.zeros(2 * len(s), dtype=torch.int32, device="cuda")
    ctx.model_pairs(d_s, len(s), d_p)
    got = d_p.cpu().numpy().view(np.uint32)
    bad = np.nonzero(got != exp)[0]
    assert bad.size == 0, f"{kind} n={n}: {bad.size} mismatches, first at {bad[:6]} got {got[bad[:3]]} exp {exp[bad[:3]]}"


def test_the_encoder_arena_grows_when_a_denser_block_arrives(oracle):
    """Round 4: the model / rANS buffers are sized from the block's own RLE0 symbol count (one 8-byte read back), the arena first for
    text-like data (0.55 symbols per byte).  A context that has only seen text meets random bytes (one symbol per byte): the arena
    grows, the stage starts over, the bytes are exact; text afterwards is exact too, and so is a block whose chunks differ wildly
    in density (zeros | random | text)."""
    import torch
    import jampack_amd as jam
    c = jam.Context(0, torch.cuda.current_stream().cuda_stream)
    try:
        n = 3 << 20
        # (images of the BWT stage, as the encoder sees them: a text block's image has ~0.4 RLE0 symbols per byte, random bytes 1.0)
        bwt_text = oracle.bwt_forward(jam.corpus.make("text", n, 1))
        mixed = np.concatenate([np.zeros(1 << 20, np.uint8), jam.corpus.make("random", (1 << 20) + 4097, 3), bwt_text[: (1 << 20) - 4097]])
        seq = [("text", bwt_text), ("random", jam.corpus.make("random", n, 2)), ("text again", oracle.bwt_forward(jam.corpus.make("text_survey", n, 5))), ("mixed", mixed)]
        ws = []
        for name, img in seq:
            want = oracle.ans_encode(img.copy())
            d_in = _dev(img)
            cap = jam.ans_capacity(len(img))
            d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
            m = c.ans_encode(d_in, len(img), d_out, cap)
            got = d_out[:m].cpu().numpy()
            assert m == len(want) and np.array_equal(got, want), name
            ws.append(int(c.stats().workspace_bytes))
        assert ws[1] > ws[0], "random bytes after text: the arena must have grown"
        assert ws[0] < 75 * n, f"text needs {ws[0] / n:.1f} bytes of arena per byte: the compact layout is not in effect"      # (4 chunks for 3 MiB + 480 B: 60 n planned, + 1/8 growth margin)
    finally:
        c.close()
