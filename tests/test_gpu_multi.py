"""jpk_blocks_compress_multi / jpk_blocks_decompress_multi on the 1-GPU box (world 1; jampack.cpp:205-224, 286-317 over the GPUs of a
node): host blocks in, results gathered in block order on the root device, each equal to the single-block entry point's bytes; once
with the root's own blocks copied device-to-device and once (subprocess, JPK_MULTI_FORCE_RCCL=1) through an RCCL send/receive to
itself, which loads librccl and builds the single-process communicator.  Every device runs its blocks through the library's batch
entries (blocks in flight, grouped small blocks): the multi entry must stay within 10 % of jpk_dev_blocks_compress on the same blocks.
No multi-GPU hardware number exists; tests/test_abi_and_host.py covers the ownership rule for 2..8 devices.  -m gpu"""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

BODY = r'''
import sys
sys.path.insert(0, %r)
import numpy as np, torch
import jampack_amd as jam
from oracle.pyoracle import Oracle
o = Oracle()
dev = torch.device("cuda", 0)
spec = [("text_survey", 2_000_000), ("zero", 300_000), ("random", 70_001), ("text", 0), ("dna", 119), ("runs", 1_048_576), ("text_survey", 3_100_007)]
blocks = [jam.corpus.make(k, n, 50 + i) for i, (k, n) in enumerate(spec)]
cap = sum(jam.ans_capacity(len(b) + 480) for b in blocks)
d_out = torch.empty(cap, dtype=torch.uint8, device=dev)
off, st = jam.blocks_compress_multi(blocks, d_out, cap, 0b1)
assert st == [0] * len(blocks), st
assert off[0] == 0 and all(off[i] <= off[i + 1] for i in range(len(blocks)))
out = d_out.cpu().numpy()
for i, t in enumerate(blocks):
    want = o.ans_encode(o.bwt_forward(t, prefill=0))
    got = out[off[i]: off[i + 1]]
    assert len(got) == len(want) and np.array_equal(got, want), (i, len(got), len(want))
# too small an output buffer is reported, nothing is written past it
small = torch.zeros(1000, dtype=torch.uint8, device=dev)
try:
    jam.blocks_compress_multi(blocks, small, 1000, 0b1)
    raise SystemExit("expected a capacity error")
except jam.JampackError as e:
    assert e.status == -2, e.status
# the decompress direction: the compressed blocks (host) come back as the blocks' bytes, in block order, on the root
comp = [out[off[i]: off[i + 1]].copy() for i in range(len(blocks))]
raw = [len(b) for b in blocks]
d_back = torch.empty(sum(raw) + 64, dtype=torch.uint8, device=dev)
doff, dst = jam.blocks_decompress_multi(comp, raw, d_back, sum(raw) + 64, 0b1)
assert dst == [0] * len(blocks), dst
back = d_back.cpu().numpy()
for i, t in enumerate(blocks):
    assert doff[i + 1] - doff[i] == len(t) and np.array_equal(back[doff[i]: doff[i + 1]], t), i
# a corrupt block fails alone
bad = [c.copy() for c in comp]
bad[0][len(bad[0]) // 2] ^= 0x55
try:
    jam.blocks_decompress_multi(bad, raw, d_back, sum(raw) + 64, 0b1)
    raise SystemExit("expected an error for the corrupt block")
except jam.JampackError as e:
    assert e.status in (-3, -2), e.status
# ... and the healthy blocks are THERE (ADVICE r5: the gather used to be skipped as a whole, their ranges pointed at garbage): every block
# whose status is 0 is gathered at its range, the failed block's range is empty, the call returns the failed block's status
d_back.zero_()
boff, bst, brc = jam.blocks_decompress_multi(bad, raw, d_back, sum(raw) + 64, 0b1, check=False)
assert brc in (-3, -2) and bst[0] == brc and bst[1:] == [0] * (len(blocks) - 1), (brc, bst)
assert boff[1] - boff[0] == 0
back = d_back.cpu().numpy()
for i, t in enumerate(blocks):
    if i:
        assert boff[i + 1] - boff[i] == len(t) and np.array_equal(back[boff[i]: boff[i + 1]], t), i
# when the gather cannot run nothing is reported as done: every status is an error, every range empty
soff, sst, src = jam.blocks_compress_multi(blocks, small, 1000, 0b1, check=False)
assert src == -2 and all(x != 0 for x in sst) and soff == [0] * (len(blocks) + 1), (src, sst, soff)
# the caller's device is what it was (also behind jpk_release_idle, which destroys contexts and slabs: ADVICE r5), and a second call re-uses / re-creates the slabs
assert torch.cuda.current_device() == 0
jam.release_idle()
assert torch.cuda.current_device() == 0
off2, st2 = jam.blocks_compress_multi(blocks, d_out, cap, 0b1, 4)
assert off2 == off and st2 == st and np.array_equal(d_out.cpu().numpy()[: off[-1]], out[: off[-1]])
jam.shutdown()
print("multi ok")
'''

SPEED = r'''
import sys, time
sys.path.insert(0, %r)
import numpy as np, torch
import jampack_amd as jam
dev = torch.device("cuda", 0)
blocks = [jam.corpus.make("text_survey", 32 << 20, 70 + i) for i in range(12)]
cap = sum(jam.ans_capacity(len(b) + 480) for b in blocks)
d_out = torch.empty(cap, dtype=torch.uint8, device=dev)
ctx = jam.Context(0, torch.cuda.current_stream().cuda_stream)
d_in = [torch.from_numpy(b).to(dev) for b in blocks]
caps = [jam.ans_capacity(len(b) + 480) for b in blocks]
d_o = [torch.empty(c, dtype=torch.uint8, device=dev) for c in caps]
def t_batch():
    t0 = time.perf_counter(); n, st = ctx.blocks_compress(d_in, [len(b) for b in blocks], d_o, caps, 8); torch.cuda.synchronize(); return time.perf_counter() - t0, n
def t_multi():
    t0 = time.perf_counter(); off, st = jam.blocks_compress_multi(blocks, d_out, cap, 0b1, 8); torch.cuda.synchronize(); return time.perf_counter() - t0, off
t_batch(); t_multi()
tb = min(t_batch()[0] for _ in range(3)); n = t_batch()[1]
tm = min(t_multi()[0] for _ in range(3)); off = t_multi()[1]
tot = sum(len(b) for b in blocks)
pcie = tot / 45e9                       # the multi entry also moves the blocks in from (pageable) host memory ...
print("batch %%.1f ms = %%.0f MB/s, multi %%.1f ms = %%.0f MB/s (host copies ~%%.1f ms, under the blocks before them)" %% (tb * 1e3, tot / tb / 1e6, tm * 1e3, tot / tm / 1e6, pcie * 1e3))
assert [off[i + 1] - off[i] for i in range(len(blocks))] == list(n)
# ... block k + 1 while block k is being compressed (round 6): the WALL time of the multi entry, copies included, within 10 %% of the batch
# entry on resident inputs (round 5 subtracted an assumed PCIe time here: every copy sat in front of the first kernel)
assert tm <= 1.10 * tb + 0.005, (tm, tb)
jam.shutdown()
print("speed ok")
'''


@pytest.mark.parametrize("force_rccl", [0, 1])
def test_blocks_compress_multi_world1(force_rccl):
    env = dict(os.environ)
    env["JPK_MULTI_FORCE_RCCL"] = str(force_rccl)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-c", BODY % ROOT], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "multi ok" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


def test_multi_entry_runs_at_the_batch_entry_speed():
    """every device drives its blocks through jpk_dev_blocks_compress (8 in flight), not one block at a time, and
    moves block k + 1 in from host memory while block k is compressed: wall time within 10 % of that entry on resident inputs"""
    env = dict(os.environ)
    r = subprocess.run([sys.executable, "-c", SPEED % ROOT], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "speed ok" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])
