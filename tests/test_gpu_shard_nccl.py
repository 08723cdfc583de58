"""The gather of the compressed blocks over RCCL (torch.distributed backend "nccl"), world size 2, one process per GPU.
Runs only where at least two GPUs are visible (the 1-GPU test box skips it; the gloo tests cover the logic there).  -m gpu"""
import os

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    from jampack_amd import shard
    nblocks = 5
    rng = np.random.default_rng(5)
    payloads = [rng.integers(0, 256, 100_000 + 37_777 * b, dtype=np.uint8) for b in range(nblocks)]
    local = [torch.from_numpy(payloads[b]).to(dev) for b in shard.my_blocks(nblocks, rank, world)]
    ok = True
    for _ in range(3):                                   # repeated: the receive buffers of one gather are free for the next
        got = shard.gather_blocks(local, dst=0, device=dev, max_local=3)
        torch.cuda.current_stream().synchronize()
        if rank == 0:
            ordered = shard.assemble_in_block_order(got, nblocks)
            ok = ok and all(np.array_equal(ordered[b].cpu().numpy(), payloads[b]) for b in range(nblocks))
        else:
            ok = ok and got is None
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs")
def test_gather_blocks_over_rccl_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29950 + os.getpid() % 40
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok in res), res
