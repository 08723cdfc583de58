"""BASELINE.json configs[3]: "enwik9 as 64 MiB blocks sharded across 8 x MI355X (RCCL gather), 1/2/4/8 scaling".

On ONE GPU the ownership rule, the per-rank work and the in-order reassembly of shard.py / bench.py are exercised for
every G in {1, 2, 4, 8}: the 15 blocks of the 1 000 000 000-B enwik9-like stream (14 x 64 MiB + 60 475 904 B) are
compressed "as rank r" for r = 0..G-1 (block b belongs to rank b mod G, jampack.cpp:205-224's in-order block loop), the
per-rank results are put back in block order, and the concatenated payload must be the same bytes for every G -- and
equal to the reference build's ForwardBwt + Ans::Encode on the blocks it is checked on (first, a middle one, the short
last one).  The collective itself (all_gather of sizes + padded gather, shard.gather_blocks) runs here on the real
backend with world size 1 (nccl = RCCL); world size 2 runs on CPU under gloo (tests/test_shard_gloo.py).  No hardware
scaling curve exists until the driver has an 8-GPU node.  -m gpu"""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BS = 64 << 20


@pytest.fixture(scope="module")
def stream15():
    """the 15 blocks resident in HBM + one context per simulated lane"""
    import torch
    import jampack_amd as jam
    from jampack_amd import corpus
    ranges = corpus.block_ranges(corpus.workload_bytes("enwik9"), BS)
    assert len(ranges) == 15 and ranges[-1][1] == 60_475_904
    dev = torch.device("cuda", 0)
    d_in = []
    for (o, n) in ranges:
        d, _ = corpus.load_or_make("enwik9", start=o, count=n)
        d_in.append(torch.from_numpy(d).to(dev))
    ctxs = [jam.Context(0, None) for _ in range(2)]
    for c in ctxs:
        c.reserve(BS)
    yield torch, jam, ranges, d_in, ctxs
    for c in ctxs:
        c.close()


def _compress_as_rank(torch, jam, ctxs, d_in, mine):
    """what one rank of bench.py --workload enwik9 does with its blocks: two in flight, each on its own context"""
    import concurrent.futures as cf
    out = [None] * len(mine)

    def lane(k):
        for pos in range(k, len(mine), len(ctxs)):
            b = mine[pos]
            n = d_in[b].numel()
            cap = jam.ans_capacity(n + jam.TRAILER)
            buf = torch.empty(cap, dtype=torch.uint8, device=d_in[b].device)
            m = ctxs[k].block_compress(d_in[b], n, buf, cap)
            out[pos] = buf[:m].clone()

    with cf.ThreadPoolExecutor(max_workers=len(ctxs)) as pool:
        for f in [pool.submit(lane, k) for k in range(len(ctxs))]:
            f.result()
    return out


def test_sharded_stream_is_identical_for_every_gpu_count(stream15, ref):
    torch, jam, ranges, d_in, ctxs = stream15
    from jampack_amd import shard
    digests = {}
    payload = {}
    for G in (1, 2, 4, 8):
        per_rank = [_compress_as_rank(torch, jam, ctxs, d_in, shard.my_blocks(15, r, G)) for r in range(G)]
        ordered = shard.assemble_in_block_order(per_rank, 15)
        h = hashlib.sha256()
        for t in ordered:
            h.update(t.cpu().numpy().tobytes())
        digests[G] = h.hexdigest()
        payload[G] = [int(t.numel()) for t in ordered]
        if G == 1:
            first = ordered
    assert len(set(digests.values())) == 1, digests
    assert payload[1] == payload[8]
    # against the reference itself (ForwardBwt + Ans::Encode, full-size blocks): first, middle, short last block
    for b in (0, 7, 14):
        t = d_in[b].cpu().numpy()
        want = ref.ans_encode(ref.bwt_forward(t))
        got = first[b].cpu().numpy()
        assert len(got) == len(want) and hashlib.sha256(got.tobytes()).hexdigest() == hashlib.sha256(want.tobytes()).hexdigest(), f"block {b}"
    # and every block decodes back to its source through the GPU path
    back = torch.empty(BS, dtype=torch.uint8, device=d_in[0].device)
    for b in range(15):
        n = d_in[b].numel()
        assert ctxs[0].block_decompress(first[b], first[b].numel(), back, n) == n
        assert torch.equal(back[:n], d_in[b])


_NCCL_WORLD1 = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from jampack_amd import shard
g = torch.Generator().manual_seed(5)
blocks = [torch.randint(0, 256, (n,), dtype=torch.uint8, generator=g).to(dev) for n in (1000, 1, 70001, 4096, 12345)]
got = shard.gather_blocks(blocks, dst=0, device=dev)
assert len(got) == 1 and len(got[0]) == len(blocks) and all(torch.equal(a, b) for a, b in zip(got[0], blocks))
ordered = shard.assemble_in_block_order(got, len(blocks))
assert all(torch.equal(a, b) for a, b in zip(ordered, blocks))
empty = shard.gather_blocks([], dst=0)            # a rank that owns no block still hands RCCL a CUDA tensor
assert empty == [[]]
dist.barrier(); dist.destroy_process_group()
print("NCCL_WORLD1_OK")
"""


def test_rccl_gather_world_size_1(tmp_path):
    """shard.gather_blocks on the real backend (nccl = RCCL), one rank: the collective calls, the padding and the
    zero-block rank.  Fresh process: a process group must not leak into the other tests."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    r = subprocess.run([sys.executable, "-c", _NCCL_WORLD1, ROOT], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "NCCL_WORLD1_OK" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


def test_bench_stream_mode_with_forced_gather():
    """bench.py --workload enwik9 end to end on one GPU with the RCCL gather forced on (JPK_FORCE_GATHER): the line says
    config 4, scaling strong, and rank 0's reassembled payload matches its own blocks.  Truncated to 3 blocks to stay short."""
    env = dict(os.environ, JPK_FORCE_GATHER="1", MASTER_PORT="29534")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "enwik9", "--limit-bytes", str(2 * BS + 5_000_000),
                        "--steps", "2", "--warmup", "1", "--no-extras"], capture_output=True, text=True, timeout=1200, env=env)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["gather_ok"] is True and line["scaling"] == "strong" and line["n_gpus"] == 1
    assert line["config"]["block_bytes"] == [BS, BS, 5_000_000] and "b mod 1" in line["config"]["workload"]
    assert line["value"] > 0
