"""Real files on the GPU (the generators of jampack_amd/corpus.py model text; real trees have what models lack: licence headers repeated
thousands of times, generated tables, runs of blanks, near-duplicate files).  A 64 MiB block and a group of 8 MiB blocks out of the source
and text files of this image's Python and ROCm trees (corpus.system_sources: sorted path order, the GPU box runs the same image), compared
with the reference build's BYTES -- SHA-256 of the BWT image (bwt.cpp:22-65) and of the rANS stream (ans.cpp:113-234) -- plus the round trip.
Skipped where the trees hold less than the block.  -m gpu"""
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


@pytest.fixture(scope="module")
def sources():
    from jampack_amd import corpus
    t = corpus.system_sources(96 << 20)
    if t is None:
        pytest.skip("this image's source trees hold less than 96 MiB")
    return t


def test_64_mib_of_real_source_files_equal_the_reference(gpu, ref, sources):
    torch, jam, ctx = gpu
    t = sources[: 64 << 20]
    n = len(t)
    dev = torch.device("cuda", 0)
    d_in = torch.from_numpy(t).to(dev)
    cap = jam.ans_capacity(n + 480)
    d_bwt = torch.empty(n + 480, dtype=torch.uint8, device=dev)
    d_enc = torch.empty(cap, dtype=torch.uint8, device=dev)
    assert ctx.bwt_forward(d_in, n, d_bwt, n + 480) == n + 480
    s = ctx.stats()
    clen = ctx.ans_encode(d_bwt, n + 480, d_enc, cap)
    rb = ref.bwt_forward(t)
    assert hashlib.sha256(rb.tobytes()).hexdigest() == _sha(d_bwt)
    re_ = ref.ans_encode(rb)
    assert len(re_) == clen and hashlib.sha256(re_.tobytes()).hexdigest() == _sha(d_enc[:clen])
    d_back = torch.empty(n, dtype=torch.uint8, device=dev)
    assert ctx.block_decompress(d_enc, clen, d_back, n) == n
    assert torch.equal(d_back, d_in)
    print(f"real sources, 64 MiB: ratio {clen / n:.3f}, rounds {s.sa_rounds}, key order {s.sa_key_order}, unresolved per round {list(s.sa_round_active)[: s.sa_rounds]}")
    assert s.sa_rounds <= 24, s.sa_rounds


@pytest.mark.parametrize("skip_mib", [0, 200])
def test_64_mib_of_real_shared_libraries_equal_the_reference(gpu, ref, skip_mib):
    """machine code, gfx code objects, symbol and string tables, padding (corpus.system_binaries: the ROCm tree's shared libraries): the first
    64 MiB (ratio 0.065) and the 64 MiB from byte 200 Mi on (ratio 0.003: tables repeated almost verbatim; 16 rounds, most of the block in
    groups above 1024 members for seven of them, a pair round that resolves nothing)"""
    from jampack_amd import corpus
    torch, jam, ctx = gpu
    t = corpus.system_binaries(64 << 20, skip_mib << 20)
    if t is None:
        pytest.skip("this image's ROCm tree holds less")
    n = len(t)
    dev = torch.device("cuda", 0)
    d_in = torch.from_numpy(t).to(dev)
    cap = jam.ans_capacity(n + 480)
    d_bwt = torch.empty(n + 480, dtype=torch.uint8, device=dev)
    d_enc = torch.empty(cap, dtype=torch.uint8, device=dev)
    assert ctx.bwt_forward(d_in, n, d_bwt, n + 480) == n + 480
    s = ctx.stats()
    clen = ctx.ans_encode(d_bwt, n + 480, d_enc, cap)
    rb = ref.bwt_forward(t)
    assert hashlib.sha256(rb.tobytes()).hexdigest() == _sha(d_bwt)
    re_ = ref.ans_encode(rb)
    assert len(re_) == clen and hashlib.sha256(re_.tobytes()).hexdigest() == _sha(d_enc[:clen])
    d_back = torch.empty(n, dtype=torch.uint8, device=dev)
    assert ctx.block_decompress(d_enc, clen, d_back, n) == n
    assert torch.equal(d_back, d_in)
    print(f"real shared libraries from {skip_mib} MiB: ratio {clen / n:.3f}, rounds {s.sa_rounds}, pair rounds {[r for r in range(40) if (s.sa_pair_rounds >> r) & 1]}")


def test_group_of_real_8_mib_blocks_equals_the_reference(gpu, ref, sources):
    """four 8 MiB blocks behind the first 64 MiB (other files): one jpk_dev_blocks_compress call = one group sort"""
    torch, jam, ctx = gpu
    dev = torch.device("cuda", 0)
    blocks = [sources[(64 << 20) + k * (8 << 20): (64 << 20) + (k + 1) * (8 << 20)] for k in range(4)]
    d_in = [torch.from_numpy(np.ascontiguousarray(b)).to(dev) for b in blocks]
    caps = [jam.ans_capacity(len(b) + 480) for b in blocks]
    outs = [torch.empty(c, dtype=torch.uint8, device=dev) for c in caps]
    lens, st = ctx.blocks_compress(d_in, [len(b) for b in blocks], outs, caps, 4)
    assert st == [0] * 4
    for b, o, m in zip(blocks, outs, lens):
        want = ref.ans_encode(ref.bwt_forward(b))
        assert m == len(want) and np.array_equal(o[:m].cpu().numpy(), want)
