"""jpk_blocks_compress_multi on the 1-GPU box (world 1): host blocks in, compressed blocks gathered in block order on the root
device, each equal to the single-block entry point's bytes; once with the root's own blocks copied device-to-device and once
(subprocess, JPK_MULTI_FORCE_RCCL=1) through an RCCL send/receive to itself, which loads librccl and builds the single-process
communicator.  No multi-GPU hardware number exists; tests/test_abi_and_host.py covers the ownership rule for 2..8 devices.  -m gpu"""
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
jam.shutdown()
print("multi ok")
'''


@pytest.mark.parametrize("force_rccl", [0, 1])
def test_blocks_compress_multi_world1(force_rccl):
    env = dict(os.environ)
    env["JPK_MULTI_FORCE_RCCL"] = str(force_rccl)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-c", BODY % ROOT], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "multi ok" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])
