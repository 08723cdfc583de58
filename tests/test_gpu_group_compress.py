"""Small blocks compressed in groups (jpk_dev_blocks_compress: one suffix sort and one set of entropy grids over the blocks of a
group; bwt_fwd.hip jpk_fwd_bwt_group_device, ans_enc.hip jpk_ans_encode_group_device).

The reference's default block is 8 MiB and its smallest 1 MiB (format.hpp:20-22); Jampack::Compress feeds `Threads` of them at a
time (jampack.cpp:205-224).  Whatever the grouping, every block's bytes must be those of the reference for that block alone:
checked against the oracle per block (the reference build for the 8 MiB block when it is present) and against the single-block
entry point.  -m gpu"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

MiB = 1 << 20


@pytest.fixture(scope="module")
def gpu():
    import torch
    import jampack_amd as jam
    assert torch.cuda.is_available()
    ctx = jam.Context(0, torch.cuda.current_stream().cuda_stream)
    yield torch, jam, ctx
    ctx.close()


def _compress_all(torch, jam, ctx, blocks, in_flight=4, caps=None):
    dev = torch.device("cuda", 0)
    d_in = [torch.from_numpy(np.ascontiguousarray(b)).to(dev) if len(b) else torch.empty(1, dtype=torch.uint8, device=dev) for b in blocks]
    caps = caps or [jam.ans_capacity(len(b) + jam.TRAILER) for b in blocks]
    d_out = [torch.empty(max(c, 1), dtype=torch.uint8, device=dev) for c in caps]
    n, st = ctx.blocks_compress(d_in, [len(b) for b in blocks], d_out, caps, in_flight)
    return n, st, d_out, d_in


def test_mixed_group_equals_the_oracle_block_by_block(gpu, oracle):
    """the sizes the verdict names: 0, 1, 119, 120, 121, 1 MiB - 1, 1 MiB, 8 MiB, tails that are not multiples of 120, a zero block
    beside a text block -- one call, grouped inside the library"""
    from oracle.pyoracle import Ref
    torch, jam, ctx = gpu
    spec = [("text", 0), ("text", 1), ("dna", 119), ("two", 120), ("text", 121), ("text_survey", MiB - 1), ("text_survey", MiB), ("zero", 2 * MiB + 77),
            ("text_survey", 8 * MiB), ("runs", 700_001), ("random", 70_001), ("silesia", 3 * MiB + 61), ("geometric", 123_456), ("repeat4k", 500_000)]
    blocks = [jam.corpus.make(k, n, 31 + i) for i, (k, n) in enumerate(spec)]
    n, st, d_out, _ = _compress_all(torch, jam, ctx, blocks)
    assert st == [0] * len(blocks)
    big = Ref() if Ref.available() else oracle
    for i, t in enumerate(blocks):
        impl = big if len(t) > 4 * MiB else oracle
        want = impl.ans_encode(impl.bwt_forward(t)) if len(t) >= 120 else oracle.ans_encode(oracle.bwt_forward(t, prefill=0))
        got = d_out[i][: n[i]].cpu().numpy()
        assert n[i] == len(want) and np.array_equal(got, want), f"block {i} {spec[i]}: {n[i]} vs {len(want)} bytes"


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_groups_equal_single_block_calls(gpu, seed):
    """random mixes of small blocks (and one large one between them, which splits the groups): grouped == one block at a time"""
    torch, jam, ctx = gpu
    rng = np.random.default_rng(seed)
    kinds = ["text_survey", "text", "zero", "runs", "dna", "random", "two", "repeat4k", "geometric", "samples16"]
    blocks = []
    for i in range(24):
        n = int(rng.choice([0, 7, 119, 120, 4096, 65_537, 300_000, MiB, MiB + 1, 2 * MiB + 500, 3_000_000]))
        blocks.append(jam.corpus.make(kinds[int(rng.integers(len(kinds)))], n, 100 * seed + i))
    blocks.insert(9, jam.corpus.make("text_survey", 17 * MiB + 5, 7))          # larger than the grouping limit: goes alone
    n, st, d_out, d_in = _compress_all(torch, jam, ctx, blocks, in_flight=3)
    assert st == [0] * len(blocks)
    dev = torch.device("cuda", 0)
    for i, t in enumerate(blocks):
        cap = jam.ans_capacity(len(t) + jam.TRAILER)
        one = torch.empty(cap, dtype=torch.uint8, device=dev)
        m = ctx.block_compress(d_in[i], len(t), one, cap)
        assert m == n[i] and torch.equal(one[:m], d_out[i][:m]), f"block {i} ({len(t)} B)"
    # and the streams decode to the inputs through the batch decoder
    backs = [torch.empty(max(len(t), 1), dtype=torch.uint8, device=dev) for t in blocks]
    bn, bst = ctx.blocks_decompress([d_out[i][: n[i]] for i in range(len(blocks))], n, backs, [len(t) for t in blocks])
    assert bst == [0] * len(blocks)
    for i, t in enumerate(blocks):
        assert bn[i] == len(t) and np.array_equal(backs[i][: len(t)].cpu().numpy(), t), i


def test_a_block_that_does_not_fit_is_reported_alone(gpu):
    torch, jam, ctx = gpu
    blocks = [jam.corpus.make("text_survey", 600_000, 5), jam.corpus.make("random", 400_000, 6), jam.corpus.make("text_survey", 700_000, 7)]
    caps = [jam.ans_capacity(len(b) + jam.TRAILER) for b in blocks]
    good_n, good_st, good_out, _ = _compress_all(torch, jam, ctx, blocks)
    assert good_st == [0, 0, 0]
    caps[1] = 1000                                   # random bytes do not shrink: block 1 cannot fit
    n, st, d_out, _ = _compress_all(torch, jam, ctx, blocks, caps=caps)
    assert st[0] == 0 and st[2] == 0 and st[1] == -2 and n[1] == 0
    for i in (0, 2):
        assert n[i] == good_n[i] and torch.equal(d_out[i][: n[i]], good_out[i][: n[i]])


def test_many_one_mib_blocks_through_one_call(gpu, oracle):
    """the reference's smallest block size: 40 x 1 MiB of one text stream, several groups in flight; a sample against the oracle"""
    torch, jam, ctx = gpu
    data = jam.corpus.make("text_survey", 40 * MiB, 12)
    blocks = [data[k * MiB: (k + 1) * MiB] for k in range(40)]
    n, st, d_out, _ = _compress_all(torch, jam, ctx, blocks, in_flight=8)
    assert st == [0] * 40
    for i in (0, 13, 39):
        want = oracle.ans_encode(oracle.bwt_forward(blocks[i]))
        assert n[i] == len(want) and np.array_equal(d_out[i][: n[i]].cpu().numpy(), want), i


_HOOK_CHILD = r"""
import sys
sys.path.insert(0, %r)
import numpy as np, torch
import jampack_amd as jam
from jampack_amd._lib import lib
dev = torch.device("cuda", 0)
ctx = jam.Context(0, torch.cuda.current_stream().cuda_stream)
blocks = [jam.corpus.make(k, n, 9 + i) for i, (k, n) in enumerate((("text_survey", 1 << 20), ("random", 300_000), ("dna", 2_000_001), ("zero", 70_000), ("text", 1_500_000)))]
d_in = [torch.from_numpy(b).to(dev) for b in blocks]
caps = [jam.ans_capacity(len(b) + 480) for b in blocks]
def run(caps_now):
    d_out = [torch.zeros(max(c, 1), dtype=torch.uint8, device=dev) for c in caps_now]
    n, st = ctx.blocks_compress(d_in, [len(b) for b in blocks], d_out, caps_now, 2)
    return n, st, [d_out[i][: n[i]].cpu().numpy() for i in range(len(blocks))], d_out
n0, st0, ref, _ = run(caps)
assert st0 == [0] * len(blocks)
# 1. the whole group fails before it runs (as if its arena could not be had): every block comes back through the single-block path
assert lib().jpk_debug_group_fail_next(1) == 0
n1, st1, got, _ = run(caps)
assert st1 == [0] * len(blocks) and list(n1) == list(n0) and all(np.array_equal(a, b) for a, b in zip(ref, got)), (st1, n1, n0)
# 2. one block's buffer is too small: that block alone says JPK_E_CAPACITY, nothing is written to its buffer, the others are complete
small = list(caps); small[2] = 1000
n2, st2, got2, d_out2 = run(small)
assert st2 == [0, 0, -2, 0, 0], st2
assert int(d_out2[2].max().item()) == 0
for i in (0, 1, 3, 4):
    assert n2[i] == n0[i] and np.array_equal(got2[i], ref[i]), i
print("GROUP_HOOK_OK")
"""


def test_a_failed_group_is_retried_block_by_block_and_a_small_buffer_fails_alone():
    """a group-level failure used to fail every block of the group (up to 256); the decode combiner already had this fallback"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, JPK_DEBUG_HOOKS="1")
    r = subprocess.run([sys.executable, "-c", _HOOK_CHILD % root], env=env, capture_output=True, text=True, timeout=600)
    assert "GROUP_HOOK_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
