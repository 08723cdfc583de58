"""Multi-rank path on CPU: world_size 2, gloo.  Covers block ownership and the variable-size output gather that
bench.py uses with RCCL on the GPU box."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, port, q, nblocks=5):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from jampack_amd import shard
    mine = shard.my_blocks(nblocks, rank, world)
    rng = np.random.default_rng(100)
    payloads = [rng.integers(0, 256, 1000 + 377 * b, dtype=np.uint8) for b in range(nblocks)]
    local = [torch.from_numpy(payloads[b]) for b in mine]
    got = shard.gather_blocks(local, dst=0)
    ok = True
    if rank == 0:
        for r in range(world):
            owned = shard.my_blocks(nblocks, r, world)
            ok = ok and len(got[r]) == len(owned)
            for t, b in zip(got[r], owned):
                ok = ok and np.array_equal(t.numpy(), payloads[b])
        # file order (jampack.cpp:220-224): block b comes back at index b whatever the ownership was
        ordered = shard.assemble_in_block_order(got, nblocks)
        ok = ok and len(ordered) == nblocks and all(np.array_equal(ordered[b].numpy(), payloads[b]) for b in range(nblocks))
    else:
        ok = got is None
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("nblocks", [5, 1, 0])      # 1 and 0: a rank (or every rank) owns no block at all
def test_gather_blocks_world2(nblocks):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() + 7 * nblocks) % 200
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, nblocks)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok in res), res


def test_block_ownership_is_a_partition():
    from jampack_amd import shard
    for world in (1, 2, 4, 8):
        seen = []
        for r in range(world):
            seen += shard.my_blocks(15, r, world)
        assert sorted(seen) == list(range(15))


def _worker_mismatch(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from jampack_amd import shard
    local = [torch.zeros(10, dtype=torch.uint8) for _ in range(3 if rank == 1 else 1)]     # rank 1 owns more than max_local = 2
    try:
        shard.gather_blocks(local, dst=0, max_local=2)
        q.put((rank, "no error"))
    except ValueError as e:
        q.put((rank, "raised" if "every rank" in str(e) else str(e)))
    dist.barrier()
    dist.destroy_process_group()


def test_a_max_local_violation_raises_on_every_rank_instead_of_hanging():
    """a rank that owns more blocks than max_local used to raise alone while the others entered the all_gather"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29850 + os.getpid() % 100
    procs = [ctx.Process(target=_worker_mismatch, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, "raised"), (1, "raised")], res
