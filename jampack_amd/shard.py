"""Block sharding across the GPUs of one node (one process per GPU, torch.distributed; backend "nccl" = RCCL
over xGMI on the GPU box, "gloo" in the CPU tests).

Blocks are independent units (jampack.cpp:215-219, 313-317: one Jampack instance per OpenMP thread and block), so
the data path has no collective.  The only exchange is the final gather of the per-block outputs on one rank, which
the reference does implicitly by writing blocks to the file in order (jampack.cpp:220-224).
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def owner_of(block_index: int, world: int) -> int:
    """block b -> rank b mod world (round robin, the analogue of jampack.cpp:209-219's thread loop)"""
    return block_index % world


def my_blocks(nblocks: int, rank: int, world: int) -> list[int]:
    return [b for b in range(nblocks) if owner_of(b, world) == rank]


def _collective_device(local, device, group) -> torch.device:
    """device of the collective's tensors: explicit > the local blocks' > what the backend needs (a rank that owns no
    block -- fewer blocks than ranks -- must still hand RCCL a CUDA tensor)"""
    if device is not None:
        return torch.device(device)
    if local:
        return local[0].device
    if dist.get_backend(group) == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def gather_blocks(local: list[torch.Tensor], dst: int = 0, group=None, device=None):
    """Variable-size gather of 1-D uint8 tensors (one per local block) onto rank `dst`.

    Returns on dst a list (per rank) of lists of tensors in local block order, elsewhere None.
    Two collectives: all_gather of the size vectors, then one gather of a flat buffer padded to the largest rank
    total (compressed blocks are ~0.15-0.25 x the input, so the padding is noise next to the compute).
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = _collective_device(local, device, group)
    nloc = torch.tensor([len(local)], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(nloc) for _ in range(world)]
    dist.all_gather(counts, nloc, group=group)
    maxn = int(max(int(c.item()) for c in counts))
    sizes = torch.zeros(max(maxn, 1), dtype=torch.int64, device=dev)
    for i, t in enumerate(local):
        sizes[i] = t.numel()
    allsizes = [torch.zeros_like(sizes) for _ in range(world)]
    dist.all_gather(allsizes, sizes, group=group)
    totals = [int(s.sum().item()) for s in allsizes]
    pad = max(max(totals), 1)
    flat = torch.zeros(pad, dtype=torch.uint8, device=dev)
    off = 0
    for t in local:
        flat[off:off + t.numel()] = t
        off += t.numel()
    bufs = [torch.empty(pad, dtype=torch.uint8, device=dev) for _ in range(world)] if rank == dst else None
    dist.gather(flat, bufs, dst=dst, group=group)
    if rank != dst:
        return None
    out = []
    for r in range(world):
        n = int(counts[r].item())
        o = 0
        blocks = []
        for i in range(n):
            sz = int(allsizes[r][i].item())
            blocks.append(bufs[r][o:o + sz].clone())
            o += sz
        out.append(blocks)
    return out


def assemble_in_block_order(per_rank: list[list[torch.Tensor]], nblocks: int) -> list[torch.Tensor]:
    """undo owner_of(): per_rank[r][k] is the k-th block owned by rank r = block r + k * world -> blocks 0..nblocks-1 in file
    order (what CompWriteBlock's in-order loop produces, jampack.cpp:220-224)"""
    world = len(per_rank)
    out = []
    for b in range(nblocks):
        r = owner_of(b, world)
        out.append(per_rank[r][b // world])
    return out
