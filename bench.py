#!/usr/bin/env python3
"""bench.py -- block hot path throughput on MI355X (see DESIGN.md section "Measurement").

  python bench.py --gpus N --steps K --warmup W

A "step" = one pass of the compress hot path (forward BWT -> rANS encode, Jampack::Comp() tail, jampack.cpp:40-41)
over one batch = the enwik8-like workload (100 000 000 B, BASELINE.json configs[1]) cut into 64 MiB blocks
(67 108 864 + 32 891 136 B), inputs already resident in HBM.  For N > 1 every rank compresses its own batch
(blocks are independent, jampack.cpp:215: weak scaling) and the compressed blocks are gathered on rank 0 with RCCL.
value = uncompressed bytes of all ranks / max-over-ranks time, in MB/s (1e6 B/s).

Extra keys on the same JSON line: `decompress` (rANS decode -> inverse BWT over the same batch), per-stage
timings, `roofline` for the dominant kernel (HIP-event timed inside the library on the launch stream) and
`cpu_baseline` (the real reference, oracle/_ref, on this box's host cores; rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# blocks in flight live on separate HIP streams; the ROCm default of 4 hardware queues serialises them beyond two
# (tools/dec_scaling.py).  Must be set before the HIP runtime initialises (torch initialises it before our library loads).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

import numpy as np  # noqa: E402


# algorithmic HBM bytes per processed unit of every timed kernel class
ALG_BYTES_PER_UNIT = {
    "k_rs_hist": (8, "sorted (key,value) pair"),
    "k_rs_scatter": (24, "sorted (key,value) pair"),
    "k_scan_*": (12, "u32 element"),
    "k_init_keys/k_make_keys/k_win_heads": (12, "suffix"),
    "k_seg_round": (28, "active suffix"),
    "sa rerank kernels": (16, "suffix"),
    "k_bwt_gather": (6, "block byte"),
    "k_enc_hist/k_enc_prep": (1, "block byte"),
    "k_enc_mtf": (2, "block byte"),
    "k_rle_*": (2, "block byte"),
    "k_cls_*/k_quasi_build": (9, "RLE0 symbol"),
    "k_adaptive": (15, "RLE0 symbol"),
    "k_pairs": (44, "RLE0 symbol"),
    "k_rans_lanes": (20, "rANS pair"),
    "k_emit_scan/k_put_*": (13, "rANS pair"),
}


def pmc_traffic(kernel_class: str):
    """HBM bytes per launch of a kernel class from the committed rocprofv3 PMC passes (profiles/r01_pmc_traffic.json:
    separate --pmc FETCH_SIZE / --pmc WRITE_SIZE runs of this same command; KiB counters; reads x2 for the gfx950
    half-count of wide coalesced loads, MI355X_MICROARCH.md section HBM).  None if no PMC summary is committed."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    if not os.path.exists(path):
        return None
    tab = json.load(open(path))
    names = [n.strip().rstrip("*") for n in kernel_class.split("/")]
    fetch = write = launches = 0.0
    for k, v in tab.items():
        if any(k.startswith(n) for n in names if n.startswith("k_")):
            fetch += 2.0 * v["fetch_KiB_raw"] * 1024
            write += v["write_KiB"] * 1024
            launches += v["launches"]
    return round((fetch + write) / launches) if launches else None


FORCE_GATHER = bool(int(os.environ.get("JPK_FORCE_GATHER", "0")))      # exercise the RCCL gather with WORLD_SIZE=1 (test hook)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="enwik8", choices=["enwik6", "enwik8", "enwik9", "silesia"])
    ap.add_argument("--block-mib", type=int, default=64)
    ap.add_argument("--limit-bytes", type=int, default=0, help="truncate the workload (debug only; marks the line invalid)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--contexts", type=int, default=2, help="blocks in flight per GPU (one context + HIP stream each)")
    ap.add_argument("--cpu-sample-mib", type=int, default=24)
    return ap.parse_args()


def cpu_baseline(block: np.ndarray, sample_mib: int):
    """the reference itself (oracle/_ref/libjamref.so) on the host cores, bounded sample of the same block"""
    from oracle.pyoracle import Oracle, Ref
    cores = os.cpu_count() or 1
    n = min(len(block), sample_mib << 20)
    n -= n % 120
    sample = np.ascontiguousarray(block[:n])
    kind = "reference" if Ref.available() else "port"
    impl = Ref() if kind == "reference" else Oracle()
    t0 = time.perf_counter()
    bwt = impl.bwt_forward(sample)
    t1 = time.perf_counter()
    enc = impl.ans_encode(bwt)
    t2 = time.perf_counter()
    if kind == "reference":
        dec = impl.ans_decode(enc, len(bwt), threads=cores)
        t3 = time.perf_counter()
        back = impl.bwt_inverse(dec, threads=cores)
    else:
        dec = impl.ans_decode(enc, len(bwt))
        t3 = time.perf_counter()
        back = impl.bwt_inverse(dec)
    t4 = time.perf_counter()
    assert np.array_equal(back, sample)
    mb = n / 1e6
    return {
        "value": round(mb / (t2 - t0), 3), "unit": "MB/s", "cores": cores if kind == "reference" else 1, "kind": kind,
        "sample": f"first {n} B of block 0 as one block: ForwardBwt (divsufsort, OpenMP {cores} threads) + Ans::Encode (single-threaded by design)",
        "forward_bwt_MBps": round(mb / (t1 - t0), 3), "ans_encode_MBps": round(mb / (t2 - t1), 3),
        "decompress_MBps": round(mb / (t4 - t2), 3), "ans_decode_MBps": round(mb / (t3 - t2), 3), "inverse_bwt_MBps": round(mb / (t4 - t3), 3),
    }, enc


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1 or FORCE_GATHER:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import jampack_amd as jam
    from jampack_amd import corpus

    # ---- workload: every rank gets its own batch of the same shape (weak scaling over independent blocks) ----
    data, source = corpus.load_or_make(args.workload, limit=args.limit_bytes or None, seed_offset=1000 * rank)
    blocks = corpus.split_blocks(data, args.block_mib << 20)
    batch_bytes = int(sum(len(b) for b in blocks))
    d_in = [torch.from_numpy(np.ascontiguousarray(b)).to(dev) for b in blocks]
    caps = [jam.ans_capacity(len(b) + jam.TRAILER) for b in blocks]
    d_out = [torch.empty(c, dtype=torch.uint8, device=dev) for c in caps]
    stream = torch.cuda.current_stream()
    ctx = jam.Context(local_rank, stream.cuda_stream)           # stage breakdown / profiling context (torch's stream)
    # blocks are independent (jampack.cpp:215: one Jampack instance per OpenMP thread): keep `--contexts` of them in
    # flight, each on its own context = own HBM arena + own HIP stream, driven by one host thread each
    import concurrent.futures as cf
    nctx = max(1, min(args.contexts, len(blocks)))
    ctxs = [jam.Context(local_rank, None) for _ in range(nctx)]
    for c in [ctx] + ctxs:
        c.reserve(max(len(b) for b in blocks))          # HBM arenas sized before anything is timed
    pool = cf.ThreadPoolExecutor(max_workers=nctx)
    order = sorted(range(len(blocks)), key=lambda i: -len(blocks[i]))          # largest first
    lanes = [order[k::nctx] for k in range(nctx)]

    sizes = [0] * len(blocks)
    from jampack_amd import shard
    gathered = [None]

    def lane_work(k):
        for i in lanes[k]:
            sizes[i] = ctxs[k].block_compress(d_in[i], len(blocks[i]), d_out[i], caps[i])

    def compress_step():
        for f in [pool.submit(lane_work, k) for k in range(nctx)]:
            f.result()
        if world > 1 or FORCE_GATHER:
            # the only exchange of the path: the compressed blocks of every rank -> rank 0 (all_gather of the sizes, one
            # gather of a buffer padded to the largest rank total), RCCL over xGMI; jampack_amd/shard.py
            gathered[0] = shard.gather_blocks([d_out[i][: sizes[i]] for i in range(len(blocks))], dst=0)

    def sync_all():
        if world > 1 or FORCE_GATHER:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        compress_step()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        compress_step()
    sync_all()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ms_per_step = dt / args.steps * 1e3
    value = world * batch_bytes / 1e6 / (dt / args.steps)

    # ---- un-timed extras on rank 0: stage breakdown, decompress leg, parity flags, roofline ----
    extra = {}
    if rank == 0:
        st = ctx.stats()
        ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731  (same stream as the library's launches)
        stage_ms = {"forward_bwt": 0.0, "ans_encode": 0.0, "ans_decode": 0.0, "inverse_bwt": 0.0}
        comp_sizes = []
        ok = True
        for i, b in enumerate(blocks):
            n = len(b)
            d_bwt = torch.empty(n + jam.TRAILER, dtype=torch.uint8, device=dev)
            d_enc = torch.empty(caps[i], dtype=torch.uint8, device=dev)
            d_dec = torch.empty(n + jam.TRAILER, dtype=torch.uint8, device=dev)
            d_back = torch.empty(max(n, 1), dtype=torch.uint8, device=dev)
            e = [ev() for _ in range(5)]
            e[0].record(stream)
            ctx.bwt_forward(d_in[i], n, d_bwt, n + jam.TRAILER)
            e[1].record(stream)
            clen = ctx.ans_encode(d_bwt, n + jam.TRAILER, d_enc, caps[i])
            e[2].record(stream)
            dl = ctx.ans_decode(d_enc, clen, d_dec, n + jam.TRAILER)
            e[3].record(stream)
            bl = ctx.bwt_inverse(d_dec, dl, d_back, n)
            e[4].record(stream)
            torch.cuda.synchronize()
            for k, name in enumerate(stage_ms):
                stage_ms[name] += e[k].elapsed_time(e[k + 1])
            comp_sizes.append(clen)
            ok = ok and bl == n and bool(torch.equal(d_back[:n], d_in[i])) and clen == sizes[i] and bool(torch.equal(d_enc[:clen], d_out[i][:clen]))
            del d_bwt, d_enc, d_dec, d_back
        mb = batch_bytes / 1e6
        extra["stages_ms"] = {k: round(v, 3) for k, v in stage_ms.items()}
        extra["stages_MBps"] = {k: round(mb / (v / 1e3), 1) for k, v in stage_ms.items() if v > 0}
        # decompress leg (rANS decode -> inverse BWT, jampack.cpp:49-50) over the same batch, blocks in flight like compress
        d_cmp = [d_out[i][: sizes[i]].clone() for i in range(len(blocks))]
        d_dcm = [torch.empty(max(len(b), 1), dtype=torch.uint8, device=dev) for b in blocks]
        dsz = [0] * len(blocks)

        def dlane(k):
            for i in lanes[k]:
                dsz[i] = ctxs[k].block_decompress(d_cmp[i], sizes[i], d_dcm[i], len(blocks[i]))

        def decompress_step():
            for f in [pool.submit(dlane, k) for k in range(nctx)]:
                f.result()

        decompress_step()
        torch.cuda.synchronize()
        td0 = time.perf_counter()
        decompress_step()
        torch.cuda.synchronize()
        td = time.perf_counter() - td0
        ok = ok and all(dsz[i] == len(blocks[i]) and bool(torch.equal(d_dcm[i][: len(blocks[i])], d_in[i])) for i in range(len(blocks)))
        extra["decompress"] = {"value": round(mb / td, 1), "unit": "MB/s", "ms_per_step": round(td * 1e3, 3),
                               "one_block_at_a_time_MBps": round(mb / ((stage_ms["ans_decode"] + stage_ms["inverse_bwt"]) / 1e3), 1),
                               "inverse_bwt_MBps": round(mb / (stage_ms["inverse_bwt"] / 1e3), 1)}
        extra["round_trip_ok"] = ok
        # the same passes as a continuous stream: 4 passes over the batch through the same contexts WITHOUT a barrier
        # between passes (a free context takes the next block, as the reference's OpenMP block loop over a long file
        # does, jampack.cpp:215).  Informational: `value` above stays the barrier-per-step number.
        import queue
        import threading
        for mode, fn, args_of in (("compress", "block_compress", lambda i: (d_in[i], len(blocks[i]), d_out[i], caps[i])),
                                  ("decompress", "block_decompress", lambda i: (d_cmp[i], sizes[i], d_dcm[i], len(blocks[i])))):
            passes = 4
            tasks = queue.Queue()
            for _ in range(passes):
                for i in order:
                    tasks.put(i)

            def drain(k):
                while True:
                    try:
                        i = tasks.get_nowait()
                    except queue.Empty:
                        return
                    getattr(ctxs[k], fn)(*args_of(i))

            torch.cuda.synchronize()
            ts0 = time.perf_counter()
            th = [threading.Thread(target=drain, args=(k,)) for k in range(nctx)]
            for t_ in th:
                t_.start()
            for t_ in th:
                t_.join()
            torch.cuda.synchronize()
            tsd = time.perf_counter() - ts0
            extra.setdefault("streamed", {})[mode] = {"value": round(passes * mb / tsd, 1), "unit": "MB/s", "passes": passes,
                                                      "ms_per_pass": round(tsd / passes * 1e3, 3)}
        del d_cmp, d_dcm
        extra["compressed_bytes"] = int(sum(comp_sizes))
        if gathered[0] is not None:          # rank 0 holds every rank's compressed blocks: check its own against the source
            extra["gather_ok"] = all(bool(torch.equal(gathered[0][0][i], d_out[i][: sizes[i]])) for i in range(len(blocks))) and len(gathered[0]) == world
        extra["sa_rounds_last_block"] = int(st.sa_rounds)
        extra["workspace_bytes"] = int(ctx.stats().workspace_bytes)
        # algorithmic HBM traffic of the stages: fwd BWT 10 B/B, ANS 4+c B/B, inverse BWT 12 B/B
        c = sum(comp_sizes) / batch_bytes
        alg = {"forward_bwt": 10.0, "ans_encode": 4.0 + c, "ans_decode": 4.0 + c, "inverse_bwt": 12.0}
        extra["stages_alg_GBps"] = {k: round(alg[k] * batch_bytes / 1e9 / (stage_ms[k] / 1e3), 2) for k in alg if stage_ms[k] > 0}
        # per-kernel HIP-event timing (events recorded by the library on the launch stream) of one more compress pass
        ctx.profile_enable(2)
        for i, b in enumerate(blocks):
            ctx.block_compress(d_in[i], len(b), d_out[i], caps[i])
        tab = ctx.profile_table()
        ctx.profile_enable(0)
        rows = []
        for r in tab:
            bpu = ALG_BYTES_PER_UNIT.get(r["name"])
            if not bpu or not r["units"] or r["ms"] <= 0:
                continue
            ach = bpu[0] * r["units"] / 1e9 / (r["ms"] / 1e3)
            rows.append({"kernel": r["name"], "ms_total": round(r["ms"], 3), "launches": r["launches"], "avg_launch_us": round(r["ms"] * 1e3 / r["launches"], 2),
                         "alg_bytes_per_unit": bpu[0], "unit_is": bpu[1], "units": r["units"], "achieved": round(ach, 2), "frac": round(ach / 8000.0, 5)})
        rows.sort(key=lambda r: -r["ms_total"])
        if rows:
            d0 = rows[0]
            extra["roofline"] = {"bound": "hbm", "achieved": d0["achieved"], "peak": 8000.0, "unit": "GB/s", "frac": d0["frac"],
                                 "traffic": pmc_traffic(d0["kernel"]),
                                 "kernel": d0["kernel"], "avg_launch_us": d0["avg_launch_us"], "launches": d0["launches"],
                                 "alg_bytes_per_launch": round(d0["alg_bytes_per_unit"] * d0["units"] / d0["launches"]),
                                 "note": "dominant kernel of the compress pass by total time; achieved = algorithmic bytes per launch / mean launch time (HIP events on the launch stream); traffic = PMC bytes per launch from profiles/r01_pmc_traffic.json; this kernel is bound by dependent-instruction latency (4 serial rANS chains per chunk), not by HBM"}
            extra["roofline_kernels"] = rows[:8]
        if world == 1 and not args.no_cpu_baseline:
            try:
                cb, ref_enc = cpu_baseline(blocks[0], args.cpu_sample_mib)
                n = min(len(blocks[0]), args.cpu_sample_mib << 20)
                n -= n % 120
                # bit-exact vs the CPU reference on the same sample block
                d_s = d_in[0][:n].contiguous()
                d_o = torch.empty(jam.ans_capacity(n + jam.TRAILER), dtype=torch.uint8, device=dev)
                m = ctx.block_compress(d_s, n, d_o, d_o.numel())
                cb["bit_exact_vs_cpu"] = bool(m == len(ref_enc) and np.array_equal(d_o[:m].cpu().numpy(), ref_enc))
                extra["cpu_baseline"] = cb
            except Exception as ex:  # the baseline must never take the GPU line down
                extra["cpu_baseline"] = {"value": None, "unit": "MB/s", "cores": 0, "kind": "error", "sample": repr(ex)}

    if rank == 0:
        line = {
            "metric": "MB/s compress (forward BWT + rANS encode) on 64 MiB blocks; bit-exact vs CPU ref",
            "value": round(value, 2), "unit": "MB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8/int32", "data": "synthetic" if source == "synthetic" else source,
            "config": {"workload": f"{args.workload}-like {batch_bytes} B per GPU as {args.block_mib} MiB blocks ({len(blocks)} blocks), forward BWT + rANS encode, inputs resident in HBM"
                       + ("" if not args.limit_bytes else " [TRUNCATED: not a valid headline]"),
                       "block_bytes": [len(b) for b in blocks], "parallelism": (f"blocks sharded over {world} GPU(s), RCCL gather of compressed blocks" if world > 1 else "1 GPU") + f", {nctx} blocks in flight per GPU"},
        }
        line.update(extra)
        print(json.dumps(line), flush=True)
    pool.shutdown()
    for c in ctxs:
        c.close()
    ctx.close()
    if world > 1 or FORCE_GATHER:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
