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
