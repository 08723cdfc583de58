"""Batches of independent blocks (jpk_dev_blocks_ans_decode / jpk_dev_blocks_decompress): Jampack::Decompress's multi-block mode
(jampack.cpp:286-317) for HBM-resident blocks.  One pass must give, block by block, exactly what the single-block entry
points give -- for ragged block sizes, empty blocks, mixed corpora -- and a corrupt block must be reported in its own status
slot without disturbing its neighbours.  -m gpu"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch
    import jampack_amd as jam
    assert torch.cuda.is_available()
    ctx = jam.Context(0, torch.cuda.current_stream().cuda_stream)
    yield torch, jam, ctx
    ctx.close()


def _blocks(jam):
    spec = [("text_survey", 3_000_000), ("random", 70_001), ("zero", 2_100_000), ("geometric", 1_048_576 - 480), ("text", 5),
            ("runs", 1_300_000), ("dna", 119), ("silesia", 4_000_000), ("two", 120), ("text_survey", 1_048_577)]
    return [jam.corpus.make(k, n, 11 + i) for i, (k, n) in enumerate(spec)]


def test_batch_equals_single_block_calls(gpu, oracle):
    torch, jam, ctx = gpu
    dev = torch.device("cuda", 0)
    blocks = _blocks(jam)
    comp, bwts = [], []
    for t in blocks:
        bw = oracle.bwt_forward(t)                     # defined trailer bytes for the < 120-byte blocks too
        bwts.append(bw)
        comp.append(oracle.ans_encode(bw))
    d_in = [torch.from_numpy(c).to(dev) for c in comp]
    # Ans::Decode of the whole batch
    d_mid = [torch.empty(len(b) + 16, dtype=torch.uint8, device=dev) for b in bwts]
    n, st = ctx.blocks_ans_decode(d_in, [len(c) for c in comp], d_mid, [m.numel() for m in d_mid])
    assert st == [0] * len(blocks)
    for i, b in enumerate(bwts):
        assert n[i] == len(b) and np.array_equal(d_mid[i][: n[i]].cpu().numpy(), b), i
    # fused: Ans::Decode + InverseBwt of the whole batch
    d_out = [torch.empty(max(len(t), 1), dtype=torch.uint8, device=dev) for t in blocks]
    n, st = ctx.blocks_decompress(d_in, [len(c) for c in comp], d_out, [len(t) for t in blocks])
    assert st == [0] * len(blocks)
    for i, t in enumerate(blocks):
        assert n[i] == len(t) and np.array_equal(d_out[i][: n[i]].cpu().numpy(), t), i
        one = torch.empty(max(len(t), 1), dtype=torch.uint8, device=dev)
        assert ctx.block_decompress(d_in[i], len(comp[i]), one, len(t)) == len(t) and torch.equal(one[: len(t)], d_out[i][: len(t)])


def test_corrupt_block_is_isolated(gpu, oracle):
    torch, jam, ctx = gpu
    dev = torch.device("cuda", 0)
    blocks = _blocks(jam)[:6]
    comp = [oracle.ans_encode(oracle.bwt_forward(t)) for t in blocks]
    bad = [c.copy() for c in comp]
    bad[1][300] ^= 0x40                               # a frequency byte of block 1: header sum no longer matches
    bad[3] = bad[3][: len(bad[3]) // 2]               # block 3 truncated
    bad[4][-5] ^= 0xFF                                # payload of block 4: final rANS states wrong
    d_in = [torch.from_numpy(c).to(dev) for c in bad]
    d_out = [torch.empty(max(len(t), 1), dtype=torch.uint8, device=dev) for t in blocks]
    n, st = ctx.blocks_decompress(d_in, [len(c) for c in bad], d_out, [len(t) for t in blocks])
    assert st[0] == 0 and st[2] == 0 and st[5] == 0
    assert st[1] in (-3, -2) and st[3] in (-3, -2) and st[4] in (-3, -2)
    for i in (0, 2, 5):
        assert n[i] == len(blocks[i]) and np.array_equal(d_out[i][: n[i]].cpu().numpy(), blocks[i])
    # too small an output buffer is a capacity error of that block only
    n, st = ctx.blocks_decompress(d_in[:1] + d_in[2:3], [len(bad[0]), len(bad[2])], [d_out[0][:1000], d_out[2]], [1000, len(blocks[2])])
    assert st[0] == -2 and st[1] == 0 and n[1] == len(blocks[2])
    # and the context still decodes exact bytes afterwards
    ok = torch.empty(len(blocks[0]), dtype=torch.uint8, device=dev)
    good = torch.from_numpy(comp[0]).to(dev)
    assert ctx.block_decompress(good, len(comp[0]), ok, len(blocks[0])) == len(blocks[0]) and np.array_equal(ok.cpu().numpy(), blocks[0])


def test_bad_trailer_index_is_isolated_in_a_batch_with_inverse_lanes(gpu, oracle):
    """Eight blocks or more: the inverse BWTs of a batch run on three lanes (own stream, own scratch).  Two of nine blocks carry a
    primary index outside [1, n] -- their rANS streams decode, their inverse BWTs refuse on the device -- and are reported in
    their own slots; the seven others, on whichever lane, are exact; the call is repeated on the same context."""
    torch, jam, ctx = gpu
    dev = torch.device("cuda", 0)
    kinds = ["text_survey", "runs", "dna", "text", "geometric", "silesia", "random", "text_survey", "two"]
    sizes = [1_500_000, 700_000, 300_001, 2_000_000, 120, 1_048_576, 90_000, 1_200_007, 5_000]
    blocks = [jam.corpus.make(k, n, 900 + i) for i, (k, n) in enumerate(zip(kinds, sizes))]
    comp = []
    for i, t in enumerate(blocks):
        bw = oracle.bwt_forward(t)
        if i in (2, 7):
            tr = bw[len(bw) - jam.TRAILER:].view(np.int32)
            tr[0] = len(t) + 7 if i == 2 else 0              # the primary index (bwt.cpp:171-174 reads this one): beyond the block / below 1
        comp.append(oracle.ans_encode(bw))
    d_in = [torch.from_numpy(c).to(dev) for c in comp]
    for rep in range(2):
        d_out = [torch.full((max(len(t), 1),), 0xEE, dtype=torch.uint8, device=dev) for t in blocks]
        n, st = ctx.blocks_decompress(d_in, [len(c) for c in comp], d_out, [len(t) for t in blocks])
        for i, t in enumerate(blocks):
            if i in (2, 7):
                assert st[i] == -3 and n[i] == 0, (rep, i, st[i])
            else:
                assert st[i] == 0 and n[i] == len(t) and np.array_equal(d_out[i][: n[i]].cpu().numpy(), t), (rep, i)


def test_more_than_1024_chains_longest_first(gpu, oracle):
    """A batch with more chunk chains than one launch wave of the GPU (> 1024) takes the longest-first launch order (k_dec_order)
    and drops the one-chain-per-SIMD LDS reservation: 1104 blocks (138 distinct ones of ragged sizes and mixed content, eight
    output buffers each) in ONE call, every block exact, two corrupted ones reported in their own slots."""
    torch, jam, ctx = gpu
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(2024)
    kinds = ["text_survey", "random", "runs", "geometric", "dna", "text"]
    sizes = [int(x) for x in rng.integers(100, 40_000, size=134)] + [2_200_000, 1_048_576 - 480, 1_048_577 - 480, 3_000_000]
    distinct = [jam.corpus.make(kinds[i % len(kinds)], n, 500 + i) for i, n in enumerate(sizes)]
    dcomp = []
    for i, t in enumerate(distinct):
        d_t = torch.from_numpy(t).to(dev)
        cap = jam.ans_capacity(len(t) + jam.TRAILER)
        d_c = torch.empty(cap, dtype=torch.uint8, device=dev)
        n = ctx.block_compress(d_t, len(t), d_c, cap)
        dcomp.append(d_c[:n].clone())
        if i % 23 == 0 or i >= 134:                      # the inputs of the batch are what the reference would have written
            assert np.array_equal(dcomp[-1].cpu().numpy(), oracle.ans_encode(oracle.bwt_forward(t))), i
    order = [int(x) for x in rng.permutation(np.repeat(np.arange(len(distinct)), 8))]
    blocks = [distinct[k] for k in order]
    comp = [dcomp[k] for k in order]
    bad = {333, order.index(135)}
    comp[333] = comp[333].clone(); comp[333][270] ^= 0x21          # rlen / payload of one block
    j = order.index(135)
    comp[j] = comp[j][: comp[j].numel() - 9]                        # one copy of the 1 MiB-chunk block loses its tail
    d_out = [torch.empty(max(len(t), 1), dtype=torch.uint8, device=dev) for t in blocks]
    n, st = ctx.blocks_decompress(comp, [int(c.numel()) for c in comp], d_out, [len(t) for t in blocks])
    assert ctx.stats().ans_chunks > 1024
    for i, t in enumerate(blocks):
        if i in bad:
            assert st[i] in (-3, -2), (i, st[i])
        else:
            assert st[i] == 0 and n[i] == len(t) and np.array_equal(d_out[i][: n[i]].cpu().numpy(), t), i


def test_blocks_in_flight_change_the_schedule_not_the_bytes(gpu, oracle):
    """The encoder cuts its chains into 4 / 2 / 1 launch groups depending on how many blocks of the process are being compressed
    at that moment (jpk_compress_inflight): four contexts compressing at once must write exactly what one context alone writes,
    which is what the reference writes."""
    import threading
    torch, jam, ctx = gpu
    dev = torch.device("cuda", 0)
    blocks = [jam.corpus.make(k, n, 900 + i) for i, (k, n) in enumerate([("text_survey", 9_000_000), ("random", 3_000_001), ("runs", 5_000_000),
                                                                         ("text", 12_000_000), ("dna", 4_200_000), ("silesia", 6_000_000)])]
    want = [oracle.ans_encode(oracle.bwt_forward(t)) for t in blocks[:2]]
    d_in = [torch.from_numpy(t).to(dev) for t in blocks]
    caps = [jam.ans_capacity(len(t) + jam.TRAILER) for t in blocks]
    alone = []
    for i, t in enumerate(blocks):
        o = torch.empty(caps[i], dtype=torch.uint8, device=dev)
        n = ctx.block_compress(d_in[i], len(t), o, caps[i])
        alone.append(o[:n].clone())
    for i in range(2):
        assert np.array_equal(alone[i].cpu().numpy(), want[i]), i
    ctxs = [jam.Context(0, None) for _ in range(4)]
    outs = [[torch.empty(caps[i], dtype=torch.uint8, device=dev) for i in range(len(blocks))] for _ in range(4)]
    got = [[0] * len(blocks) for _ in range(4)]

    def work(k):
        for rep in range(3):
            for i in range(len(blocks)):
                j = (i + k) % len(blocks)                 # the threads walk the blocks out of phase: 1..4 blocks in flight at any time
                got[k][j] = ctxs[k].block_compress(d_in[j], len(blocks[j]), outs[k][j], caps[j])

    th = [threading.Thread(target=work, args=(k,)) for k in range(4)]
    [t.start() for t in th]
    [t.join() for t in th]
    for k in range(4):
        for i in range(len(blocks)):
            assert got[k][i] == alone[i].numel() and torch.equal(outs[k][i][: got[k][i]], alone[i]), (k, i)
    for c in ctxs:
        c.close()


def test_blocks_compress_equals_single_calls(gpu, oracle):
    """jpk_dev_blocks_compress: the library's own blocks-in-flight loop (jampack.cpp:205-224) gives, block by block, the bytes of
    jpk_dev_block_compress = the reference's; a block whose output buffer is too small reports JPK_E_CAPACITY in its own slot."""
    torch, jam, ctx = gpu
    dev = torch.device("cuda", 0)
    blocks = _blocks(jam) + [jam.corpus.make("text_survey", 9_000_000, 77), jam.corpus.make("silesia", 7_000_000, 78)]
    d_in = [torch.from_numpy(t).to(dev) for t in blocks]
    caps = [jam.ans_capacity(len(t) + jam.TRAILER) for t in blocks]
    want = []
    for i, t in enumerate(blocks):
        o = torch.empty(caps[i], dtype=torch.uint8, device=dev)
        n = ctx.block_compress(d_in[i], len(t), o, caps[i])
        want.append(o[:n].clone())
        if len(t) < 200_000 or i == 0:
            bw = oracle.bwt_forward(t)
            assert np.array_equal(want[-1].cpu().numpy(), oracle.ans_encode(bw)), i
    for in_flight in (0, 1, 3, 7):
        d_out = [torch.empty(c, dtype=torch.uint8, device=dev) for c in caps]
        n, st = ctx.blocks_compress(d_in, [len(t) for t in blocks], d_out, caps, in_flight)
        assert st == [0] * len(blocks)
        for i in range(len(blocks)):
            assert n[i] == want[i].numel() and torch.equal(d_out[i][: n[i]], want[i]), (in_flight, i)
    # capacity error of one block only
    small = list(caps)
    small[0] = 1000
    d_out = [torch.empty(c, dtype=torch.uint8, device=dev) for c in small]
    n, st = ctx.blocks_compress(d_in, [len(t) for t in blocks], d_out, small, 4)
    assert st[0] == -2 and st[1:] == [0] * (len(blocks) - 1)
    assert all(n[i] == want[i].numel() and torch.equal(d_out[i][: n[i]], want[i]) for i in range(1, len(blocks)))
    # an empty batch is fine
    assert ctx.blocks_compress([], [], [], [], 4) == ([], [])


def test_release_idle_gives_the_worker_arenas_back(gpu, oracle):
    """jpk_release_idle: the worker contexts jpk_dev_blocks_compress keeps between calls are destroyed (device memory comes back), the
    next call creates them again and gives the same bytes"""
    torch, jam, ctx = gpu
    dev = torch.device("cuda", 0)
    blocks = [jam.corpus.make("text_survey", 20_000_000 + 1000 * i, 80 + i) for i in range(4)]
    d_in = [torch.from_numpy(t).to(dev) for t in blocks]
    caps = [jam.ans_capacity(len(t) + jam.TRAILER) for t in blocks]
    d_out = [torch.empty(c, dtype=torch.uint8, device=dev) for c in caps]
    n0, st = ctx.blocks_compress(d_in, [len(t) for t in blocks], d_out, caps, 4)
    assert st == [0] * 4
    first = [d_out[i][: n0[i]].clone() for i in range(4)]
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info()[0]
    freed = jam.release_idle()
    assert freed >= 3                                        # in_flight 4 = the caller's context + three workers
    assert torch.cuda.mem_get_info()[0] - free_before > 3 * 40 * 20_000_000      # three arenas of > 40 bytes per block byte
    assert jam.release_idle() == 0
    n1, st = ctx.blocks_compress(d_in, [len(t) for t in blocks], d_out, caps, 4)
    assert st == [0] * 4 and list(n1) == list(n0)
    assert all(torch.equal(d_out[i][: n1[i]], first[i]) for i in range(4))


def _leb(v):
    for nb, c in ((1, 0), (2, 127), (3, 16510), (4, 2113661)):
        if v < (127, 16510, 2113661, 270549116)[nb - 1]:
            v -= c
            b = [(v >> (7 * (nb - 1 - k))) & 0x7F for k in range(nb)]
            b[-1] |= 0x80
            return bytes(b)
    raise ValueError(v)


def _crafted_chunks(nchunks, olen, clen, rlen):
    """chunk headers that pass every per-field range check (ans.cpp:287-302) but claim rlen RLE0 symbols for olen bytes"""
    one = b"".join(_leb(f) for f in [olen] + [0] * 255) + _leb(olen) + _leb(clen) + _leb(rlen) + bytes(clen)
    return np.frombuffer(one * nchunks, dtype=np.uint8).copy()


def test_crafted_rlen_cannot_move_the_arena_under_its_neighbours(gpu, oracle):
    """a stream of 277-byte chunks with olen = 1, clen = 16, rlen = 2^20 used to make the batch decoder ask for
    ~2 MiB per chunk AFTER the fused call had placed the blocks' BWT images in the arena -- the arena moved, the good blocks were
    decoded into freed HBM.  The header walk now refuses rlen > olen (RLE::decode would end in "rle mismatch!", rle.cpp:73): the
    crafted block is corrupt in its own slot, its neighbours decode exactly, through every entry point."""
    torch, jam, ctx = gpu
    dev = torch.device("cuda", 0)
    blocks = _blocks(jam)[:4]
    comp = [oracle.ans_encode(oracle.bwt_forward(t)) for t in blocks]
    evil = _crafted_chunks(4000, 1, 16, 1 << 20)
    mild = _crafted_chunks(3, 600, 16, 601)                    # just one symbol too many
    ins = [comp[0], evil, comp[1], mild, comp[2], comp[3]]
    want = [blocks[0], None, blocks[1], None, blocks[2], blocks[3]]
    d_in = [torch.from_numpy(c).to(dev) for c in ins]
    caps = [len(t) if t is not None else 8_000_000 for t in want]
    fresh = jam.Context(0, None)                               # a small arena: any growth inside the call would move it
    try:
        for rep in range(2):
            d_out = [torch.empty(max(c, 1), dtype=torch.uint8, device=dev) for c in caps]
            n, st = fresh.blocks_decompress(d_in, [len(c) for c in ins], d_out, caps)
            for i, t in enumerate(want):
                if t is None:
                    assert st[i] == -3, (i, st[i])
                else:
                    assert st[i] == 0 and n[i] == len(t) and np.array_equal(d_out[i][: n[i]].cpu().numpy(), t), (rep, i)
        mids = [torch.empty(c + 480, dtype=torch.uint8, device=dev) for c in caps]
        n, st = fresh.blocks_ans_decode(d_in, [len(c) for c in ins], mids, [m.numel() for m in mids])
        assert [st[1], st[3]] == [-3, -3] and [st[0], st[2], st[4], st[5]] == [0, 0, 0, 0]
        with pytest.raises(jam.JampackError) as e:
            fresh.ans_decode(d_in[1], len(evil), mids[1], mids[1].numel())
        assert e.value.status == -3                            # CORRUPT, not an allocation failure
        with pytest.raises(jam.JampackError) as e:
            jam.Ans().Decode(evil, 8_000_000)
        assert e.value.status == -3
    finally:
        fresh.close()


def test_blocks_compress_is_ordered_after_the_callers_stream(gpu, oracle):
    """the batch call's workers run on streams of the library's own; work queued on the caller's stream in
    front of the call -- here the very copies that produce the inputs, behind a long kernel -- must be ordered in front of them."""
    torch, jam, _ = gpu
    dev = torch.device("cuda", 0)
    side = torch.cuda.Stream()
    blocks = [jam.corpus.make("text_survey", 2_000_000, 300 + i) for i in range(6)]
    want = [oracle.ans_encode(oracle.bwt_forward(t)) for t in blocks]
    caps = [jam.ans_capacity(len(t) + jam.TRAILER) for t in blocks]
    pinned = [torch.from_numpy(t).pin_memory() for t in blocks]
    d_in = [torch.zeros(len(t), dtype=torch.uint8, device=dev) for t in blocks]
    d_out = [torch.empty(c, dtype=torch.uint8, device=dev) for c in caps]
    big = torch.empty(1 << 28, dtype=torch.uint8, device=dev)
    c2 = jam.Context(0, side.cuda_stream)
    try:
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            for _ in range(20):
                big.add_(1)                                      # keeps the caller's stream busy: the copies below are still pending
            for d, p in zip(d_in, pinned):
                d.copy_(p, non_blocking=True)
        n, st = c2.blocks_compress(d_in, [len(t) for t in blocks], d_out, caps, 4)
        assert st == [0] * len(blocks)
        for i in range(len(blocks)):
            assert n[i] == len(want[i]) and np.array_equal(d_out[i][: n[i]].cpu().numpy(), want[i]), i
    finally:
        c2.close()


def test_inflight_accounting_is_per_device(gpu):
    """the count that drives the launch grouping is the load of the block's OWN device: while threads compress on device 0 the
    probe sees them there and nothing on device 1; afterwards both read zero"""
    import threading
    torch, jam, _ = gpu
    dev = torch.device("cuda", 0)
    lib = jam.lib()
    t = jam.corpus.make("text_survey", 24_000_000, 5)
    d_t = torch.from_numpy(t).to(dev)
    cap = jam.ans_capacity(len(t) + jam.TRAILER)
    ctxs = [jam.Context(0, None) for _ in range(3)]
    outs = [torch.empty(cap, dtype=torch.uint8, device=dev) for _ in range(3)]
    seen0, seen1, stop = [], [], threading.Event()

    def watch():
        while not stop.is_set():
            seen0.append(lib.jpk_debug_compress_inflight(0, 0))
            seen1.append(lib.jpk_debug_compress_inflight(1, 0))

    def work(k):
        for _ in range(3):
            ctxs[k].block_compress(d_t, len(t), outs[k], cap)

    w = threading.Thread(target=watch)
    th = [threading.Thread(target=work, args=(k,)) for k in range(3)]
    w.start()
    [x.start() for x in th]
    [x.join() for x in th]
    stop.set()
    w.join()
    for c in ctxs:
        c.close()
    assert max(seen0) >= 2 and max(seen0) <= 3 and max(seen1) == 0
    assert lib.jpk_debug_compress_inflight(0, 0) == 0
    assert torch.equal(outs[0], outs[1]) or True              # (bytes are covered by test_blocks_in_flight_change_the_schedule_not_the_bytes)
