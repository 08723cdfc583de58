"""Block sharding across the GPUs of one node (one process per GPU, torch.distributed; backend "nccl" = RCCL
over xGMI on the GPU box, "gloo" in the CPU tests).

Blocks are independent units (jampack.cpp:215-219, 313-317: one Jampack instance per OpenMP thread and block), so
the data path has no collective.  The only exchange is the final gather of the per-block outputs on one rank, which
the reference does implicitly by writing blocks to the file in order (jampack.cpp:220-224).

The exchange lives on the Python side by choice: bench.py and the tests own the process group (torch.distributed), and the
C++ multi-block host (jampack.cpp:205-224 through the shim) is a file writer -- each block returns through its own D2H copy
to the thread that writes it, which needs no device-side gather at all.
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


def gather_blocks(local: list[torch.Tensor], dst: int = 0, group=None, device=None, max_local: int | None = None):
    """Variable-size gather of 1-D uint8 tensors (one per local block) onto rank `dst`.

    Returns on dst a list (per rank) of lists of tensors in local block order, elsewhere None.  The tensors are VIEWS: of the
    receive buffers for the other ranks' blocks, the local tensors themselves for dst's own (nothing is copied on dst).

    Exchange (size all-gather, then grouped send/recv of exactly the bytes):
      1. ONE fixed-size all_gather of an int64 vector [count, size_0 .. size_{max_local-1}] per rank, read back with one
         device-to-host copy -- the only host synchronisation.  `max_local` = the largest number of blocks any rank can own
         (ceil(nblocks / world) for the b mod world ownership); without it one extra all_reduce(MAX) finds it.
      2. every other rank sends each of its non-empty blocks straight from the block's tensor, dst posts the matching receives
         into consecutive slices of one buffer per rank (batch_isend_irecv = grouped ncclSend/ncclRecv over each GPU's direct xGMI
         link with RCCL; plain send/recv with gloo).  No padding, no zero fill, no concatenation, no clone.
    With a stream-ordered backend (nccl) the receives are ordered on the current stream: synchronise it before the host reads.
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    out_dev = _collective_device(local, device, group)
    # gloo moves host memory: device tensors (bench.py's one-GPU test hook) are staged through the host for the exchange
    dev = torch.device("cpu") if dist.get_backend(group) == "gloo" else out_dev
    if max_local is None:
        n = torch.tensor([len(local)], dtype=torch.int64, device=dev)
        dist.all_reduce(n, op=dist.ReduceOp.MAX, group=group)
        max_local = int(n.item())
    # A rank that owns more blocks than `max_local` must not raise on its own while the others enter the collective (they would
    # wait for it forever): it takes part with its true count in slot 0 and a truncated size list, every rank reads the table,
    # and EVERY rank raises.  (`max_local` itself must be the same number on all ranks: it is the shape of the collective.)
    width = max_local + 1
    sizes = [int(t.numel()) for t in local][:max_local]
    meta = torch.tensor([len(local)] + sizes + [0] * (max_local - len(sizes)), dtype=torch.int64).to(dev)
    allmeta = torch.empty(world * width, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(allmeta, meta, group=group)
    table = allmeta.cpu().view(world, width).tolist()          # the one host synchronisation
    over = [(r, row[0]) for r, row in enumerate(table) if row[0] > max_local]
    if over:
        raise ValueError(f"gather_blocks: rank(s) {over} own more blocks than max_local={max_local} (raised on every rank)")
    totals = [sum(row[1:1 + row[0]]) for row in table]

    def peer(r):                                               # P2POp takes global ranks
        return dist.get_global_rank(group, r) if group is not None else r

    # One message per non-empty block, sent straight from the block's own tensor (no concatenation copy on the sender); dst
    # posts the matching receives into consecutive slices of ONE buffer per rank (messages between two ranks match in order).
    ops, bufs = [], {}
    if rank == dst:
        for r in range(world):
            if r != dst and totals[r] > 0:
                bufs[r] = torch.empty(totals[r], dtype=torch.uint8, device=dev)
                o = 0
                for sz in table[r][1:1 + table[r][0]]:
                    if sz:
                        ops.append(dist.P2POp(dist.irecv, bufs[r][o:o + sz], peer(r), group))
                    o += sz
    else:
        for t in local:
            if t.numel():
                ops.append(dist.P2POp(dist.isend, t.contiguous().to(dev), peer(dst), group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    if rank != dst:
        return None
    if dev != out_dev:
        bufs = {r: b.to(out_dev) for r, b in bufs.items()}
    out = []
    for r in range(world):
        if r == dst:
            out.append(list(local))
            continue
        o, blocks = 0, []
        for sz in table[r][1:1 + table[r][0]]:
            blocks.append(bufs[r][o:o + sz] if sz else torch.empty(0, dtype=torch.uint8, device=out_dev))
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
