"""GPU primitives behind the hot path (scan, radix sort, suffix array) through the C ABI probes.  -m gpu"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import torch
    from jampack_amd import Context
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    c = Context(0, torch.cuda.current_stream().cuda_stream)
    yield c
    c.close()


def _dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("n", [1, 63, 64, 100, 4095, 4096, 4097, 70000, 1_000_003, 5_000_000])
def test_exclusive_scan(ctx, n):
    rng = np.random.default_rng(n)
    a = rng.integers(0, 1000, n, dtype=np.uint32)
    d = _dev(a.view(np.int32))
    tot = ctx.exclusive_scan_u32(d, n)
    got = d.cpu().numpy().view(np.uint32)
    exp = np.concatenate(([0], np.cumsum(a, dtype=np.uint64)[:-1])).astype(np.uint32)
    assert tot == int(a.sum(dtype=np.uint64) & 0xFFFFFFFF)
    bad = np.nonzero(got != exp)[0]
    assert bad.size == 0, f"first mismatch at {bad[:5]}: got {got[bad[:5]]} exp {exp[bad[:5]]}"


@pytest.mark.parametrize("n,bits", [(1, 64), (100, 64), (4096, 64), (4097, 16), (100_000, 64), (1_000_000, 40), (3_000_001, 64)])
def test_radix_sort_pairs_is_stable_and_sorted(ctx, n, bits):
    rng = np.random.default_rng(n + bits)
    keys = rng.integers(0, 2 ** 63, n, dtype=np.uint64)
    if bits < 64:
        keys &= np.uint64((1 << bits) - 1)
    keys[: n // 3] = keys[n // 2: n // 2 + n // 3]          # plenty of duplicates: stability matters
    vals = np.arange(n, dtype=np.uint32)
    dk, dv = _dev(keys.view(np.int64)), _dev(vals.view(np.int32))
    ctx.sort_pairs_u64(dk, dv, n, 0, bits)
    gk = dk.cpu().numpy().view(np.uint64)
    gv = dv.cpu().numpy().view(np.uint32)
    order = np.argsort(keys, kind="stable")
    bad = np.nonzero(gv != order.astype(np.uint32))[0]
    assert bad.size == 0, f"first mismatch at {bad[:5]}: got {gv[bad[:5]]} exp {order[bad[:5]]}"
    assert np.array_equal(gk, keys[order])


@pytest.mark.parametrize("kind", ["text", "two", "zero", "random", "repeat4k", "dna", "runs"])
@pytest.mark.parametrize("n", [1, 2, 7, 8, 100, 5000, 200_000])
def test_suffix_array_equals_oracle(ctx, oracle, kind, n):
    import torch
    from jampack_amd import corpus
    t = corpus.make(kind, n, 9)
    dt = _dev(t)
    dsa = torch.zeros(n, dtype=torch.int32, device="cuda")
    ctx.suffix_array(dt, n, dsa)
    got = dsa.cpu().numpy()
    exp = oracle.suffix_array(t)
    bad = np.nonzero(got != exp)[0]
    assert bad.size == 0, f"{kind} n={n}: first mismatch at {bad[:5]}: got {got[bad[:5]]} exp {exp[bad[:5]]} rounds={ctx.stats().sa_rounds}"


def test_suffix_array_with_embedded_zero_bytes(ctx, oracle):
    import torch
    rng = np.random.default_rng(3)
    t = rng.integers(0, 2, 50_000, dtype=np.uint8)          # only 0x00 / 0x01: short suffixes tie on padded bytes
    t[-9:] = 0
    dt = _dev(t)
    dsa = torch.zeros(len(t), dtype=torch.int32, device="cuda")
    ctx.suffix_array(dt, len(t), dsa)
    assert np.array_equal(dsa.cpu().numpy(), oracle.suffix_array(t))


@pytest.mark.parametrize("onesweep", ["0", "1"])
def test_forward_bwt_with_either_radix_form(onesweep):
    """round 0's radix passes: the one-pass form (decoupled look-back; the default since the packed keys) and the two-pass form
    (JPK_ONESWEEP=0: histogram + scan + scatter per pass) give the reference's images, single blocks and grouped small blocks (eight
    passes).  The switch is read once per process, so each form runs in a child."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    body = (
        "import sys; sys.path.insert(0, %r)\n"
        "import numpy as np, torch, jampack_amd as jam\n"
        "from oracle.pyoracle import Oracle\n"
        "o = Oracle()\n"
        "for kind, n in (('text_survey', 1_000_000), ('zero', 300_000), ('random', 70_001), ('dna', 500_000), ('text', 4096 * 5 + 17), ('two', 4095), ('runs', 200_000)):\n"
        "    t = jam.corpus.make(kind, n, 9)\n"
        "    assert np.array_equal(jam.Bwt().ForwardBwt(t), o.bwt_forward(t)), (kind, n)\n"
        "dev = torch.device('cuda', 0); ctx = jam.Context(0, None)\n"
        "blocks = [jam.corpus.make(['text_survey', 'dna', 'zero', 'random', 'runs'][i %% 5], 50_000 + 33_333 * i, i) for i in range(12)]\n"
        "d_in = [torch.from_numpy(b).to(dev) for b in blocks]\n"
        "caps = [jam.ans_capacity(len(b) + jam.TRAILER) for b in blocks]\n"
        "d_out = [torch.empty(c, dtype=torch.uint8, device=dev) for c in caps]\n"
        "n, st = ctx.blocks_compress(d_in, [len(b) for b in blocks], d_out, caps, 2)\n"
        "assert st == [0] * 12\n"
        "for i, b in enumerate(blocks):\n"
        "    want = o.ans_encode(o.bwt_forward(b))\n"
        "    assert n[i] == len(want) and np.array_equal(d_out[i][: n[i]].cpu().numpy(), want), i\n"
        "print('radix form ok')\n") % root
    env = dict(os.environ, JPK_ONESWEEP=onesweep)
    r = subprocess.run([sys.executable, "-c", body], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "radix form ok" in r.stdout, (r.stdout[-1000:], r.stderr[-2000:])
