"""Robustness of the GPU decoders against hostile input.  Every mutated / truncated / random stream
must come back as a jpk_status -- never a fault, never a hang -- and a following VALID call on the same context must
still return exact bytes (no poisoned arena, no stuck stream).  The checks mirrored are the reference's own
(ans.cpp:91-92 invalid rANS state, ans.cpp:297-298 corrupt header, rle.cpp:72 rle mismatch, rank.cpp:104-108 invalid
frequencies, jampack.cpp:150-154 corrupt frame header, jampack.cpp:58-59 crc) -- where the reference exits, the ABI
returns JPK_E_CORRUPT / JPK_E_CAPACITY.

The loop runs in a child process under a timeout so that a fault or a hang fails this test without taking the rest of
the session (or the box) with it.  -m gpu"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_FUZZ = r"""
import sys
sys.path.insert(0, sys.argv[1])
import numpy as np
if sys.argv[2] == "batch":          # device buffers come from torch: let its HIP runtime initialise first (tests/conftest.py)
    import torch
    torch.cuda.is_available()
import jampack_amd as jam
from jampack_amd import JampackError
which, rounds = sys.argv[2], int(sys.argv[3])
import os
rng = np.random.default_rng(int(os.environ.get("JPK_FUZZ_SEED", "20261003")))     # (tools/fuzz_long.py runs other seeds)
OKSET = (-1, -2, -3)       # ARG, CAPACITY, CORRUPT are legitimate answers; DEVICE/ALLOC/NODEVICE are not

def mutate(a):
    a = a.copy()
    k = rng.integers(0, 7)
    if k == 0:                                   # flip bits
        for _ in range(int(rng.integers(1, 8))):
            a[rng.integers(0, len(a))] ^= np.uint8(1 << rng.integers(0, 8))
    elif k == 1:                                 # overwrite a run with random bytes
        o = int(rng.integers(0, len(a))); n = int(rng.integers(1, 400))
        a[o:o + n] = rng.integers(0, 256, len(a[o:o + n]), dtype=np.uint8)
    elif k == 2:                                 # truncate
        a = a[: int(rng.integers(0, len(a)))]
    elif k == 3:                                 # extend with garbage
        a = np.concatenate((a, rng.integers(0, 256, int(rng.integers(1, 3000)), dtype=np.uint8)))
    elif k == 4:                                 # header area (frequency table / LEB fields)
        for _ in range(int(rng.integers(1, 6))):
            a[rng.integers(0, min(len(a), 700))] = np.uint8(rng.integers(0, 256))
    elif k == 5:                                 # tail area (trailer indices / final states)
        for _ in range(int(rng.integers(1, 6))):
            a[len(a) - 1 - rng.integers(0, min(len(a), 480))] = np.uint8(rng.integers(0, 256))
    else:                                        # all random
        a = rng.integers(0, 256, max(1, len(a) // int(rng.integers(1, 50))), dtype=np.uint8)
    return a

kinds = ["text", "geometric", "runs", "random"]
srcs = [jam.corpus.make(k, n, 77 + i) for i, (k, n) in enumerate(zip(kinds, (300_000, 1_200_000, 150_000, 40_000)))]
bwts = [jam.Bwt().ForwardBwt(t) for t in srcs]
encs = [jam.Ans().Encode(b) for b in bwts]
frames = [jam.jam_block_write(t, 1 << 21) for t in srcs]
bad = ok = 0
if which == "batch":
    # the batch entry (jpk_dev_blocks_decompress): two hostile blocks next to two valid ones in ONE call, every round on a context
    # whose arena is only as large as the last batch needed -- a block that talks the decoder into growing the arena would move
    # it under the valid blocks' BWT images (ADVICE r2).  Valid neighbours must come back exact, hostile ones with a status.
    dev = torch.device("cuda", 0)
    ctx = jam.Context(0, None)
    comp = [jam.block_compress(t) for t in srcs]

    def header_attack(a):
        # rewrite one of the three LEB128 length fields of the first chunk header (olen / clen / rlen) to an extreme value
        a = a.copy()
        terms = np.flatnonzero(a[:1400] & 0x80)
        if len(terms) < 259:
            return mutate(a)
        f = int(rng.integers(256, 259))
        lo, hi = int(terms[f - 1]) + 1, int(terms[f]) + 1
        big = int(rng.choice([0, 1, 15, 16, 1 << 20, (1 << 20) + 1, 270549115]))
        enc = []
        for nb, c in ((1, 0), (2, 127), (3, 16510), (4, 2113661), (5, 270549116)):
            if nb == 5 or big < (127, 16510, 2113661, 270549116)[nb - 1]:
                v = big - c
                enc = [(v >> (7 * (nb - 1 - k))) & 0x7F for k in range(nb)]
                enc[-1] |= 0x80
                break
        return np.concatenate((a[:lo], np.array(enc, dtype=np.uint8), a[hi:]))

    for it in range(rounds):
        victims = set(int(x) for x in rng.choice(len(srcs), 2, replace=False))
        ins = [(header_attack(comp[i]) if rng.integers(0, 2) else mutate(comp[i])) if i in victims else comp[i] for i in range(len(srcs))]
        d_in = [torch.from_numpy(np.ascontiguousarray(c)).to(dev) for c in ins]
        d_out = [torch.empty(len(t), dtype=torch.uint8, device=dev) for t in srcs]
        n, st = ctx.blocks_decompress(d_in, [len(c) for c in ins], d_out, [len(t) for t in srcs])
        for i, t in enumerate(srcs):
            if i in victims:
                assert st[i] in OKSET + (0,), f"round {it} block {i}: status {st[i]}"
                bad += st[i] != 0; ok += st[i] == 0
            else:
                assert st[i] == 0 and n[i] == len(t) and np.array_equal(d_out[i].cpu().numpy(), t), f"round {it}: valid block {i} disturbed (status {st[i]})"
    ctx.close()
    print(f"FUZZ_OK {which} rejected={bad} accepted={ok}")
    sys.exit(0)
for it in range(rounds):
    i = it % len(srcs)
    try:
        if which == "ans":
            s = mutate(encs[i]); out = jam.Ans().Decode(s, len(bwts[i]) + int(rng.integers(-2000, 2000)))
        elif which == "bwt":
            s = mutate(bwts[i]); out = jam.Bwt().InverseBwt(s)
        else:
            s = mutate(frames[i]); out, used = jam.jam_block_read(s, 1 << 21)
        ok += 1                                  # a mutation may be harmless (or decode to other bytes under a matching crc: never for 'jam')
        if which == "jam":
            assert np.array_equal(out, srcs[i]), "a frame with a valid crc decoded to different bytes"
    except JampackError as e:
        assert e.status in OKSET, f"round {it}: status {e.status}"
        bad += 1
    if it % 10 == 9 or it == rounds - 1:         # the context must still be healthy: exact bytes for a valid call
        j = (it // 10) % len(srcs)
        assert np.array_equal(jam.Ans().Decode(encs[j], len(bwts[j])), bwts[j])
        assert np.array_equal(jam.Bwt().InverseBwt(bwts[j]), srcs[j])
        got, used = jam.jam_block_read(frames[j], 1 << 21)
        assert np.array_equal(got, srcs[j]) and used == len(frames[j])
print(f"FUZZ_OK {which} rejected={bad} accepted={ok}")
"""


@pytest.mark.parametrize("which,rounds", [("ans", 120), ("bwt", 60), ("jam", 60), ("batch", 60)])
def test_gpu_decoders_reject_hostile_streams(which, rounds):
    r = subprocess.run(["timeout", "-k", "10", "900", sys.executable, "-c", _FUZZ, ROOT, which, str(rounds)], capture_output=True, text=True)
    assert r.returncode == 0 and "FUZZ_OK" in r.stdout, f"rc={r.returncode}\n{r.stdout[-1500:]}\n{r.stderr[-3000:]}"
    rejected = int(r.stdout.split("rejected=")[1].split()[0])
    assert rejected >= rounds // 3, r.stdout            # the mutations really were hostile
