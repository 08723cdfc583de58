#!/usr/bin/env python3
"""bench.py -- block hot path throughput on MI355X (see DESIGN.md section "Measurement").

  python bench.py --gpus N --steps K --warmup W [--workload enwik8|enwik9|...]

A "step" = one pass of the compress hot path (forward BWT -> rANS encode, Jampack::Comp() tail, jampack.cpp:40-41)
over one batch of 64 MiB blocks, inputs already resident in HBM.

  default (BASELINE.json configs[1], the configuration the metric is quoted on):  the enwik8-like workload
      (100 000 000 B -> blocks of 67 108 864 + 32 891 136 B).  With N > 1 every rank compresses a batch of its own
      (blocks are independent, jampack.cpp:215) and the compressed blocks are gathered on rank 0 with RCCL: weak scaling.
  --workload enwik9 (BASELINE.json configs[3]):  ONE 1 000 000 000-B stream -> 15 blocks, block b owned by rank
      b mod N (jampack_amd/shard.py), compressed blocks gathered on rank 0 in block order = the .jam payload order of
      jampack.cpp:220-224: strong scaling.

value = uncompressed bytes of the whole job / max-over-ranks time, in MB/s (1e6 B/s).  The K timed steps are K passes over the
batch fed through one queue to `--contexts` contexts (a free context takes the next block, jampack.cpp:205-224), so consecutive
passes overlap; barrier + torch.cuda.synchronize() bracket the K steps.

N > 1 without a launcher: `python bench.py --gpus N` starts N child processes itself (one per GPU, before anything touches
the GPU in the parent); under `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` it uses the ranks it
is given.  A --gpus that disagrees with WORLD_SIZE is an error, never a silent 1-GPU run.

Extra keys on the same JSON line: `decompress` (rANS decode -> inverse BWT over the same batch, passes in flight), per-stage
timings over warmed repetitions, `roofline` for the dominant kernel (HIP-event timed inside the library on the launch stream,
over passes of the timed loop's own shape), `blocks_compress_call` (the same work through ONE jpk_dev_blocks_compress call),
`join_per_step` (round 1's loop), `sa_rounds` (active suffixes per doubling round), `phrase_book_variant`, and `cpu_baseline`
(the real reference, oracle/_ref, on this box's host cores; rank 0, N = 1 only).
"""
import argparse
import json
import os
import queue
import subprocess
import threading
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# blocks in flight live on separate HIP streams; the ROCm default of 4 hardware queues serialises them beyond two
# (tools/dec_scaling.py).  Must be set before the HIP runtime initialises (torch initialises it before our library loads).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")

import numpy as np  # noqa: E402

PROFILE_ROUND = "r06"

# algorithmic HBM bytes per processed unit of every timed kernel class, and what the class is
# actually limited by ("hbm": streaming traffic; "hbm-random": 4-byte gathers/scatters, ~55 G accesses/s whatever the bytes;
# "lds": barrier-separated LDS sort passes; "issue": instruction issue of single waves walking serial chains)
ALG_BYTES_PER_UNIT = {
    "k_rs_hist/k_os_digits": (8, "sorted (key,value) pair", "hbm"),
    "k_rs_scatter/k_os_scatter": (24, "sorted (key,value) pair", "hbm"),
    "k_sym_present/k_pack_keys": (5, "suffix x launch (1 B read; 1 B read + 8 B written)", "hbm"),
    "k_lg_hist": (4, "member of a large group x pass", "hbm"),
    "k_lg_scatter": (18, "member of a large group x pass", "hbm"),
    "k_gather_win": (16, "active suffix", "hbm-random"),
    "k_seg_round": (26, "active suffix in a group <= 1024", "lds"),
    "k_r0_*/k_lg_finish/k_cmp_*": (22, "suffix", "hbm-random"),
    "k_bwt_image": (2, "block byte", "hbm"),
    "k_enc_hist/k_enc_prep": (1, "block byte", "hbm"),
    "k_enc_mtf": (2, "block byte", "issue"),
    "k_rle_*": (2, "block byte", "hbm"),
    "k_cls_*/k_quasi_build": (9, "RLE0 symbol", "hbm"),
    "k_adaptive": (22, "RLE0 symbol", "issue"),
    "k_pairs": (42, "RLE0 symbol", "hbm"),                 # round 6: no frequency sidecar (2 x 2 B per symbol less)
    "k_rans_lanes": (18.25, "rANS pair", "issue"),         # 16-byte record read; the state's low half (2 B) + 1/16 emit-mask word written (4 B of state before)
    "k_emit_*/k_put_*": (3, "rANS pair", "hbm"),           # masks 2 x 0.25 B + low halves 2 B + ~0.2 B of payload (12 B of states and frequencies read before)
}
# algorithmic bytes per block byte of the four stages (c = compressed size / block size)
STAGE_ALG = {"forward_bwt": lambda c: 10.0, "ans_encode": lambda c: 4.0 + c, "ans_decode": lambda c: 4.0 + c, "inverse_bwt": lambda c: 12.0}


def pmc_traffic(kernel_class: str, passes: float, launches: int):
    """HBM bytes per launch of a kernel class from the committed rocprofv3 PMC passes (profiles/<round>_pmc_traffic.json:
    separate --pmc FETCH_SIZE / --pmc WRITE_SIZE runs of this same command; KiB counters; reads x2 for the gfx950
    half-count of wide coalesced loads, MI355X_MICROARCH.md section HBM).  The encoder's launch shape follows the blocks in
    flight, so the file's bytes per PASS over the workload are scaled to the `launches` that `passes` passes took here.
    None if no PMC summary is committed."""
    for rnd in (PROFILE_ROUND, "r05", "r04", "r03", "r02", "r01"):
        path = os.path.join(ROOT, "profiles", f"{rnd}_pmc_traffic.json")
        if os.path.exists(path):
            break
    else:
        return None, None
    tab = json.load(open(path))
    names = [n.strip().rstrip("*") for n in kernel_class.split("/")]
    fetch = write = nl = 0.0
    for k, v in tab.items():
        if k != "_meta" and any(k.startswith(n) for n in names if n.startswith("k_")):
            fetch += 2.0 * v["fetch_KiB_raw"] * 1024
            write += v["write_KiB"] * 1024
            nl += v["launches"]
    if not nl:
        return None, os.path.basename(path)
    file_passes = tab.get("_meta", {}).get("passes")
    if file_passes and launches:
        return round((fetch + write) / file_passes * passes / launches), os.path.basename(path)
    return round((fetch + write) / nl), os.path.basename(path)


ONE_GPU_RANKS = bool(int(os.environ.get("JPK_BENCH_ONE_GPU", "0")))   # test hook: all ranks on cuda:0 over gloo (a 1-GPU box can then run the N>1 code path)
FORCE_GATHER = bool(int(os.environ.get("JPK_FORCE_GATHER", "0")))      # exercise the RCCL gather with WORLD_SIZE=1 (test hook)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="enwik8", choices=["enwik6", "enwik8", "enwik8-phrase", "enwik8-wide", "enwik9", "silesia"])
    ap.add_argument("--block-mib", type=int, default=64)
    ap.add_argument("--limit-bytes", type=int, default=0, help="truncate the workload (debug only; marks the line invalid)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="only the timed region (multi-GPU children print this anyway)")
    # 8 since round 4: a context's arena is 3.1 GB now (6 GB before), and 8 blocks in flight measured +4.8 % over 4 (+3.2 % with 6) in
    # twelve alternating runs on one box (profiles/r04_blocks_in_flight.txt)
    ap.add_argument("--contexts", type=int, default=10,
                    help="blocks in flight per GPU (one context + HIP stream each; 10 since round 6: 6 / 8 / 10 / 12 in flight measured 5 966 / 6 230 / 6 375 / 6 198 MB/s, "
                         "profiles/r06_blocks_in_flight.txt; 8 in rounds 4-5)")
    ap.add_argument("--cpu-sample-mib", type=int, default=64, help="bytes of block 0 the CPU reference is timed on")
    ap.add_argument("--no-block-sizes", action="store_true", help="skip the per_block_size extra (1 / 64 / 256 MiB blocks)")
    ap.add_argument("--master-port", type=int, default=29511)
    ap.add_argument("--child-extras", action="store_true", help=argparse.SUPPRESS)     # internal: per_block_size + host_buffers in a fresh process
    return ap.parse_args()


def launch_children(args) -> int:
    """`python bench.py --gpus N` with no launcher: one fresh child per GPU.  The parent never initialises HIP/torch.cuda
    (a process that has touched the GPU must not be replaced or forked on this pool); only rank 0 prints the JSON line."""
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(args.gpus), "MASTER_ADDR": "127.0.0.1",
                    "MASTER_PORT": str(args.master_port), "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    return rc


def cpu_model() -> str:
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(blocks, sample_mib: int):
    """The reference itself (oracle/_ref/libjamref_hot.so) on the host cores.  Sample = the first `sample_mib` MiB of
    block 0 as one block (default: the whole 64 MiB block).  Three views, as SURVEY 8d asks:
      Threads = nproc (`value`): ForwardBwt (divsufsort with OpenMP, divsufsort.cpp:1493) + Ans::Encode (single-threaded by
      design, ans.cpp:113); Threads = 1; and blocks in parallel (jampack.cpp:215: one Jampack instance per thread), which is
      the reference's own multi-core compress mode -- here every block of the batch on its own thread."""
    import threading
    from oracle.pyoracle import Oracle, Ref
    cores = os.cpu_count() or 1
    n = min(len(blocks[0]), sample_mib << 20)
    n -= n % 120
    sample = np.ascontiguousarray(blocks[0][:n])
    kind = "reference" if Ref.available() else "port"
    impl = Ref() if kind == "reference" else Oracle()
    mb = n / 1e6
    res = {"unit": "MB/s", "kind": kind, "cpu_model": cpu_model()}

    def one_pass(threads):
        if kind == "reference":
            impl.set_threads(threads)
        t0 = time.perf_counter()
        bwt = impl.bwt_forward(sample)
        t1 = time.perf_counter()
        enc = impl.ans_encode(bwt)
        t2 = time.perf_counter()
        dec = impl.ans_decode(enc, len(bwt), threads=threads) if kind == "reference" else impl.ans_decode(enc, len(bwt))
        t3 = time.perf_counter()
        back = impl.bwt_inverse(dec, threads=threads) if kind == "reference" else impl.bwt_inverse(dec)
        t4 = time.perf_counter()
        assert np.array_equal(back, sample)
        return enc, {"compress_MBps": round(mb / (t2 - t0), 3), "forward_bwt_MBps": round(mb / (t1 - t0), 3), "ans_encode_MBps": round(mb / (t2 - t1), 3),
                     "decompress_MBps": round(mb / (t4 - t2), 3), "ans_decode_MBps": round(mb / (t3 - t2), 3), "inverse_bwt_MBps": round(mb / (t4 - t3), 3)}

    enc, full = one_pass(cores if kind == "reference" else 1)
    res.update({"value": full["compress_MBps"], "cores": cores if kind == "reference" else 1})
    res.update({k: v for k, v in full.items() if k != "compress_MBps"})
    res["sample"] = (f"first {n} B of block 0 as one block; value = ForwardBwt (divsufsort, OpenMP {cores} threads) + Ans::Encode "
                     "(single-threaded by design, ans.cpp:113); threads_1 = the same with one thread; blocks_in_parallel = every block of "
                     "the batch compressed at once, one thread team each (jampack.cpp:215)")
    if kind == "reference":
        _, one = one_pass(1)
        res["threads_1"] = one
        # blocks in parallel: the reference's -t mode.  Every block of the batch (bounded to the sample size) at once.
        parts = [np.ascontiguousarray(b[: min(len(b), n) - (min(len(b), n) % 120)]) for b in blocks]
        per = max(1, cores // len(parts))

        def comp(p):
            impl.set_threads(per)
            impl.ans_encode(impl.bwt_forward(p))

        th = [threading.Thread(target=comp, args=(p,)) for p in parts]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        dt = time.perf_counter() - t0
        res["blocks_in_parallel"] = {"compress_MBps": round(sum(len(p) for p in parts) / 1e6 / dt, 3), "blocks": len(parts), "threads_per_block": per}
        try:
            res["throughput_mode"] = cpu_throughput_mode(impl, n)
        except Exception as ex:       # noqa: BLE001 -- (host memory: 64 blocks x ~0.5 GB of suffix arrays)
            res["throughput_mode"] = {"error": repr(ex)}
    return res, enc, n


def physical_cores() -> int:
    """distinct (package, core) pairs of the CPUs this process may run on; falls back to the logical count"""
    try:
        allowed = os.sched_getaffinity(0)
        seen, cpu, phys = set(), None, None
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("processor"):
                cpu = int(ln.split(":")[1])
            elif ln.startswith("physical id"):
                phys = int(ln.split(":")[1])
            elif ln.startswith("core id") and cpu in allowed:
                seen.add((phys, int(ln.split(":")[1])))
        if seen:
            return len(seen)
    except (OSError, ValueError, AttributeError):
        pass
    return os.cpu_count() or 1


def cpu_throughput_mode(impl, block_bytes: int):
    """The mode the reference itself would be run in on this box: Jampack::Compress / Decompress give every block to its own OpenMP
    thread (jampack.cpp:215, 313) -- B = min(physical cores, 64) DISTINCT blocks at once, ONE reference thread each, compress
    (ForwardBwt + Ans::Encode) and decompress (Ans::Decode + InverseBwt), aggregate MB/s of uncompressed bytes.  The blocks are
    windows of `block_bytes` at 3 MiB strides over the first 256 MiB of the enwik9-like stream (distinct blocks, one corpus
    generation).  About 10-20 s of wall time for both legs."""
    import threading
    from jampack_amd import corpus
    MiB = 1 << 20
    nb = max(1, min(physical_cores(), 64))
    # host memory: a block in flight holds its suffix array / Map (4 B per byte), its images and buffers: ~7 B per byte; use at most
    # half of what the box (or the cgroup) has free
    avail = None
    try:
        for ln in open("/proc/meminfo"):
            if ln.startswith("MemAvailable"):
                avail = int(ln.split()[1]) * 1024
        lim = open("/sys/fs/cgroup/memory.max").read().strip()
        if lim != "max":
            cur = int(open("/sys/fs/cgroup/memory.current").read())
            avail = min(avail or (1 << 62), int(lim) - cur)
    except (OSError, ValueError):
        pass
    if avail:
        nb = max(1, min(nb, int(avail // 2 // (7 * block_bytes))))
    data, _ = corpus.load_or_make("enwik9", start=0, count=256 * MiB)
    stride = min(3 * MiB, max(0, (len(data) - block_bytes)) // max(nb - 1, 1))
    stride -= stride % 120
    parts = [np.ascontiguousarray(data[k * stride: k * stride + block_bytes]) for k in range(nb)]
    impl.set_threads(1)
    encs, backs = [None] * nb, [None] * nb

    def comp(k):
        encs[k] = impl.ans_encode(impl.bwt_forward(parts[k]))

    def decomp(k):
        backs[k] = impl.bwt_inverse(impl.ans_decode(encs[k], len(parts[k]) + 480, threads=1), threads=1)

    out = {"blocks": nb, "block_bytes": int(block_bytes), "threads_per_block": 1, "physical_cores": physical_cores()}
    for name, fn in (("compress_MBps", comp), ("decompress_MBps", decomp)):
        th = [threading.Thread(target=fn, args=(k,)) for k in range(nb)]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        dt = time.perf_counter() - t0
        out[name] = round(nb * block_bytes / 1e6 / dt, 1)
        out[name.replace("_MBps", "_s")] = round(dt, 2)
    out["round_trip_ok"] = bool(all(b is not None and np.array_equal(b, p) for b, p in zip(backs, parts)))
    out["note"] = ("the reference's own multi-block mode (jampack.cpp:215, 313): one block per thread, blocks in parallel; aggregate over all blocks, "
                   "compressed and decompressed at once; windows at 3 MiB strides over the enwik9-like stream")
    return out


def host_buffers(jam, corpus):
    """PCIe-inclusive rates of the drop-in entry points (never `value`): jpk_block_compress / jpk_block_decompress through PAGEABLE
    host buffers, T threads each with a context of its own borrowed from the library (what jam_block_pipeline does through the
    shim, jampack.cpp:205-224 / 286-317), on sixteen 64 MiB blocks of the enwik9-like stream; H2D, kernels, D2H and the host copies
    all inside the timed region."""
    import threading
    MiB = 1 << 20
    bs, nb = 64 * MiB, 16
    data, _ = corpus.load_or_make("enwik9", start=0, count=256 * MiB)
    stride = (len(data) - bs) // (nb - 1)
    parts = [np.ascontiguousarray(data[k * stride: k * stride + bs]) for k in range(nb)]
    out = {"blocks": nb, "block_bytes": bs}
    comp = [None] * nb
    try:
        # un-timed: sixteen threads take a block each, so that sixteen library contexts exist with their arenas and staging buffers
        # sized (a context's first call allocates ~6 GB; threads that end return their contexts to the library's pool, the timed
        # threads below borrow them)
        def warm(k):
            jam.block_decompress(jam.block_compress(parts[k]), bs)
        th = [threading.Thread(target=warm, args=(k,)) for k in range(16)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        for T in (1, 8, 16):
            res = {}
            for leg in ("compress", "decompress"):
                back = [None] * nb
                nxt = [0]
                lock = threading.Lock()

                def work():
                    while True:
                        with lock:
                            k = nxt[0]
                            nxt[0] += 1
                        if k >= nb:
                            return
                        if leg == "compress":
                            comp[k] = jam.block_compress(parts[k])
                        else:
                            back[k] = jam.block_decompress(comp[k], bs)

                th = [threading.Thread(target=work) for _ in range(T)]
                t0 = time.perf_counter()
                for t in th:
                    t.start()
                for t in th:
                    t.join()
                dt = time.perf_counter() - t0
                res[leg + "_MBps"] = round(nb * bs / 1e6 / dt, 1)
                if leg == "decompress":
                    res["round_trip_ok"] = bool(all(np.array_equal(b, p) for b, p in zip(back, parts)))
            out[f"threads_{T}"] = res
    finally:
        jam.shutdown()
    out["note"] = ("pageable host buffers in, pageable host buffers out, every block through jpk_block_compress / jpk_block_decompress on T host threads "
                   "(one library context per thread); MB/s of uncompressed bytes, PCIe and host copies included")
    return out


def per_block_size(jam, corpus, torch, dev, device_index: int, in_flight: int):
    """BASELINE.json's metric is "per block size"; Options.BlockSize is a first-class option of the reference (format.hpp:20-22,
    main.cpp:78) and SURVEY 8d names {1, 64, 256 MiB}; 8 MiB is the reference's default.  For each size, on the first 256 MiB of the
    enwik9-like text stream:
      one_at_a_time   one block per call (jpk_dev_block_compress / _decompress), the next call starts when the last one returned;
      streamed        the blocks of a 256 / 256 / 256 / 512 MiB stream through ONE jpk_dev_blocks_compress call (small blocks are
                      compressed in groups there: one suffix sort and one set of entropy grids per group) (`in_flight` blocks in
                      flight) and ONE jpk_dev_blocks_decompress call (one grid per serial kernel over all blocks);
      same_bytes      streamed output == one-at-a-time output for every block, and every round trip == the input.
    Inputs and outputs resident in HBM; wall clock around synchronised calls; each leg warmed once."""
    MiB = 1 << 20
    data, _ = corpus.load_or_make("enwik9", start=0, count=256 * MiB)
    d_all = torch.from_numpy(np.ascontiguousarray(data)).to(dev)
    out = {}
    ctx = None
    try:
        # (block MiB, blocks of the stream, blocks timed one at a time, blocks in flight of the streamed compress: a 1 MiB block is
        # one chunk = one 5 ms chain on one of 1024 SIMDs, so small blocks want many in flight -- the library admits up to 16)
        # 8 MiB is the reference's DEFAULT_BLOCKSIZE (format.hpp:20)
        # (1 MiB and 8 MiB: both a 256 MiB stream, as Jampack::Compress would cut a 256 MiB file, jampack.cpp:205-224)
        for bm, nstream, nsingle, nfl in ((1, 256, 8, 16), (8, 32, 4, 16), (64, 4, 2, in_flight), (256, 2, 1, in_flight)):
            bs = bm * MiB
            # a fresh context per size, and the previous size's worker contexts (and their streams) released first: streams are
            # dealt onto the hardware queues round robin at creation, leftovers of a finished leg should not sit beside this one
            if ctx is not None:
                ctx.close()
            jam.shutdown()
            ctx = jam.Context(device_index, None)
            nuniq = min(nstream, (256 * MiB) // bs)
            srcs = [d_all[(k % nuniq) * bs: (k % nuniq) * bs + bs] for k in range(nstream)]
            cap = jam.ans_capacity(bs + jam.TRAILER)
            ctx.reserve(bs)
            outs = [torch.empty(cap, dtype=torch.uint8, device=dev) for _ in range(nstream)]
            backs = [torch.empty(bs, dtype=torch.uint8, device=dev) for _ in range(nstream)]
            single = torch.empty(cap, dtype=torch.uint8, device=dev)
            # one at a time
            ctx.block_compress(srcs[0], bs, single, cap)
            ctx.block_compress(srcs[0], bs, single, cap)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            sz1 = [ctx.block_compress(srcs[k], bs, outs[k], cap) for k in range(nsingle)]
            torch.cuda.synchronize()
            tc1 = (time.perf_counter() - t0) / nsingle
            ctx.block_decompress(outs[0], sz1[0], backs[0], bs)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            bk1 = [ctx.block_decompress(outs[k], sz1[k], backs[k], bs) for k in range(nsingle)]
            torch.cuda.synchronize()
            td1 = (time.perf_counter() - t0) / nsingle
            ok = all(bk1[k] == bs and bool(torch.equal(backs[k], srcs[k])) for k in range(nsingle))
            ref = [outs[k][: sz1[k]].clone() for k in range(nsingle)]
            # streamed
            ctx.blocks_compress(srcs, [bs] * nstream, outs, [cap] * nstream, nfl)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            szs, st = ctx.blocks_compress(srcs, [bs] * nstream, outs, [cap] * nstream, nfl)
            torch.cuda.synchronize()
            tcs = (time.perf_counter() - t0) / nstream
            ok = ok and st == [0] * nstream and all(szs[k] == sz1[k] and bool(torch.equal(outs[k][: szs[k]], ref[k])) for k in range(nsingle))
            ctx.blocks_decompress(outs, szs, backs, [bs] * nstream)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            bks, st = ctx.blocks_decompress(outs, szs, backs, [bs] * nstream)
            torch.cuda.synchronize()
            tds = (time.perf_counter() - t0) / nstream
            ok = ok and st == [0] * nstream and all(bks[k] == bs and bool(torch.equal(backs[k], srcs[k])) for k in range(nstream))
            mb = bs / 1e6
            out[f"{bm}MiB"] = {"block_bytes": bs, "stream_blocks": nstream, "in_flight": min(nfl, nstream),
                               "compress_MBps": {"one_at_a_time": round(mb / tc1, 1), "streamed": round(mb / tcs, 1)},
                               "decompress_MBps": {"one_at_a_time": round(mb / td1, 1), "streamed": round(mb / tds, 1)},
                               "compressed_ratio": round(sum(szs) / (bs * nstream), 4), "same_bytes": bool(ok)}
            del outs, backs, single, ref, srcs
            torch.cuda.empty_cache()
    finally:
        if ctx is not None:
            ctx.close()
        jam.shutdown()                 # the batch call's worker contexts (a 256 MiB block's arena is ~20 GB each)
    out["note"] = ("enwik9-like text; one_at_a_time = one block per call; streamed = all blocks of the stream through one jpk_dev_blocks_compress / "
                   "jpk_dev_blocks_decompress call; HBM resident; MB/s of uncompressed bytes")
    return out


def decompress_leg(jam, torch, dev, local_rank, blocks, d_in, d_cmp, sizes, nctx, reps):
    """rANS decode -> inverse BWT (jampack.cpp:49-50) over the batch: one context per block, one batched call per pass, and passes in
    flight (`JPK_BENCH_DEC_PASSES`, default 16: `value`; 4 and 8 are reported beside it).  Every output is compared with the input.  Returns (dict, ok)."""
    import queue
    import threading
    import concurrent.futures as cf
    mb = sum(len(b) for b in blocks) / 1e6
    nblk = len(blocks)
    order = sorted(range(nblk), key=lambda i: -len(blocks[i]))
    lanes = [order[k::nctx] for k in range(nctx)]
    ctxs = [jam.Context(local_rank, None) for _ in range(nctx)]
    pool = cf.ThreadPoolExecutor(max_workers=nctx)
    d_dcm = [torch.empty(max(len(b), 1), dtype=torch.uint8, device=dev) for b in blocks]
    dsz = [0] * nblk
    ok = True

    def dlane(k):
        for i in lanes[k]:
            dsz[i] = ctxs[k].block_decompress(d_cmp[i], sizes[i], d_dcm[i], len(blocks[i]))

    def decompress_step_ctx():
        for f in [pool.submit(dlane, k) for k in range(nctx)]:
            f.result()

    def decompress_step():
        # all blocks of the batch in ONE call: every serial decode kernel runs a single grid over the chunks of all blocks
        n_, st_ = ctxs[0].blocks_decompress(d_cmp, sizes, d_dcm, [len(b) for b in blocks])
        for i in range(nblk):
            dsz[i] = n_[i] if st_[i] == 0 else -1

    decompress_step_ctx()
    torch.cuda.synchronize()
    tc0 = time.perf_counter()
    for _ in range(reps):
        decompress_step_ctx()
    torch.cuda.synchronize()
    tdc = (time.perf_counter() - tc0) / reps
    decompress_step()
    torch.cuda.synchronize()
    td0 = time.perf_counter()
    for _ in range(reps):
        decompress_step()
    torch.cuda.synchronize()
    td = (time.perf_counter() - td0) / reps
    ok = ok and all(dsz[i] == len(blocks[i]) and bool(torch.equal(d_dcm[i][: len(blocks[i])], d_in[i])) for i in range(nblk))
    pool.shutdown()
    for c_ in ctxs:
        c_.close()
    # the same passes in flight, as the compress loop runs them: `ndec` contexts each take whole passes (one batched call per
    # pass) from a queue; a block is 65 serial chains, so a pass alone leaves most of the 1024 SIMDs without a chain
    ndec = int(os.environ.get("JPK_BENCH_DEC_PASSES", "16"))   # (8 until round 4: a pass is 130 serial chains on 1024 SIMDs, sixteen of them still leave every chain a SIMD of its own)
    ndec_max = max(16, ndec)
    dctxs = [jam.Context(local_rank, None) for _ in range(ndec_max)]
    dbufs = [[torch.empty(max(len(b), 1), dtype=torch.uint8, device=dev) for b in blocks] for _ in range(ndec_max)]
    dok = [True] * ndec_max

    def dec_passes(npass, nthreads):
        q_ = queue.Queue()
        for _ in range(npass):
            q_.put(1)

        def w_(k):
            while True:
                try:
                    q_.get_nowait()
                except queue.Empty:
                    return
                n_, st_ = dctxs[k].blocks_decompress(d_cmp, sizes, dbufs[k], [len(b) for b in blocks])
                dok[k] = dok[k] and all(st_[i] == 0 and n_[i] == len(blocks[i]) for i in range(nblk))

        th_ = [threading.Thread(target=w_, args=(k,)) for k in range(nthreads)]
        for t_ in th_:
            t_.start()
        for t_ in th_:
            t_.join()

    dec_passes(ndec_max, ndec_max)             # every context has decoded once (arenas sized)
    torch.cuda.synchronize()
    in_flight = {}
    for nfl in sorted({4, 8, ndec, 16}):
        npass = max(2 * nfl, reps)
        tp0 = time.perf_counter()
        dec_passes(npass, nfl)
        torch.cuda.synchronize()
        in_flight[nfl] = ((time.perf_counter() - tp0) / npass, npass)
    tdp, npass = in_flight[ndec]
    ok = ok and all(dok) and all(bool(torch.equal(dbufs[k][i][: len(blocks[i])], d_in[i])) for k in range(ndec_max) for i in range(nblk))
    for c_ in dctxs:
        c_.close()
    del dbufs
    out = {"value": round(mb / tdp, 1), "unit": "MB/s", "ms_per_step": round(tdp * 1e3, 3), "steps": npass,
           "how": f"passes over the batch fed through a queue to {ndec} contexts, one jpk_dev_blocks_decompress call (all blocks of "
                  "the batch, one grid per serial kernel) per pass; every pass verified against the input",
           "MBps_by_passes_in_flight": {str(k): round(mb / v[0], 1) for k, v in in_flight.items()},
           "one_pass_at_a_time_MBps": round(mb / td, 1), "one_pass_at_a_time_ms": round(td * 1e3, 3),
           "one_context_per_block_MBps": round(mb / tdc, 1)}
    return out, bool(ok)


def blocks_decompress_call(jam, torch, dev, d_block, d_cmp0, clen: int):
    """the decode rate a BATCH caller gets -- 64 and 128 copies of the workload's first block through ONE
    jpk_dev_blocks_decompress call each (one pass over the chunks of all blocks, the inverse BWTs on three lanes behind it), every output
    compared with the block.  The symmetric extra of `blocks_compress_call`."""
    n = int(d_block.numel())
    out = {"block_bytes": n, "how": "N copies of the workload's first block through one jpk_dev_blocks_decompress call (C ABI); the better of two calls"}
    ctx = jam.Context(0, None)
    try:
        for N in (64, 128):
            try:
                outs = [torch.empty(n, dtype=torch.uint8, device=dev) for _ in range(N)]
                encs = [d_cmp0 for _ in range(N)]
                best = None
                for _rep in range(3):                  # (the first call sizes the arena and the lanes)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    ln, st = ctx.blocks_decompress(encs, [clen] * N, outs, [n] * N)
                    torch.cuda.synchronize()
                    dt = time.perf_counter() - t0
                    if _rep:
                        best = dt if best is None else min(best, dt)
                same = st == [0] * N and all(x == n for x in ln) and all(bool(torch.equal(o, d_block)) for o in outs)
                out[str(N)] = {"value": round(N * n / 1e6 / best, 1), "unit": "MB/s", "ms_per_call": round(best * 1e3, 2), "same_bytes": bool(same)}
                del outs
                torch.cuda.empty_cache()
            except Exception as ex:       # noqa: BLE001
                out[str(N)] = {"error": repr(ex)}
    finally:
        ctx.close()
    return out


def blocks_compress_call(jam, torch, dev, local_rank, d_in, lens, caps, ref_out, ref_sizes, nctx, npl=16):
    """the library's own blocks-in-flight loop: ONE jpk_dev_blocks_compress call (C ABI) over `npl` passes' worth of the workload's blocks
    -- what a drop-in caller without threads of its own gets.  The best of three calls behind one that creates the workers' contexts and
    sizes their arenas; every output compared with `ref_out` (the same blocks compressed one at a time)."""
    nblk = len(d_in)
    l_in = [d_in[i] for _ in range(npl) for i in range(nblk)]
    l_len = [lens[i] for _ in range(npl) for i in range(nblk)]
    l_cap = [caps[i] for _ in range(npl) for i in range(nblk)]
    l_out = [torch.empty(c, dtype=torch.uint8, device=dev) for c in l_cap]
    lctx = jam.Context(local_rank, None)
    lctx.blocks_compress(l_in, l_len, l_out, l_cap, nctx)            # the workers' contexts exist and their arenas are sized
    torch.cuda.synchronize()
    tl = None
    for _rep in range(3):               # the best of three calls (until the end of round 6 one call in six of a fresh process took 4.1-4.3 GB/s where the others
                                        # took 6.1-6.5: arenas growing in the middle of a call; the workers size them up front now: profiles/r06_blocks_compress_call.txt)
        tl0 = time.perf_counter()
        ln_, ls_ = lctx.blocks_compress(l_in, l_len, l_out, l_cap, nctx)
        torch.cuda.synchronize()
        t1 = (time.perf_counter() - tl0) / npl
        tl = t1 if tl is None else min(tl, t1)
    lctx.close()
    jam.shutdown()          # the workers' contexts (and their streams) go back: whatever follows creates contexts of its own, and
                            # streams beyond the 32 hardware queues share them (a chain kernel then blocks its neighbour)
    lib_ok = ls_ == [0] * len(l_in) and all(ln_[j] == ref_sizes[j % nblk] and bool(torch.equal(l_out[j][: ln_[j]], ref_out[j % nblk][: ln_[j]])) for j in range(len(l_in)))
    mb = sum(lens) / 1e6
    return {"value": round(mb / tl, 1), "unit": "MB/s", "ms_per_pass": round(tl * 1e3, 3), "passes": npl, "in_flight": nctx, "same_bytes": bool(lib_ok),
            "how": "one jpk_dev_blocks_compress call (C ABI) over all the blocks of 16 passes"}


def child_extras(args):
    """`per_block_size` and `host_buffers` in a process of their own.  HIP deals streams onto the hardware queues round robin as they
    are created and never rebalances: after the timed loop, the decompress legs and the batch calls of this bench (~60 streams created
    and destroyed) a NEW context's stream shares a queue with one that is still alive, and everything it runs is 25-30 % slower
    (per_block_size inside the bench process: 2.7 GB/s for the 1 MiB stream; in a fresh process on the same box: 4.5;
    profiles/r04_fresh_process.txt).  A caller's process does not carry that history, so these two extras are measured the way a
    caller would see them; the parent waits, idle, and merges the JSON."""
    import torch
    import jampack_amd as jam
    from jampack_amd import corpus
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    out = {}
    try:
        # the decompress leg of the default line: the workload's blocks, compressed here first
        data, _ = corpus.load_or_make(args.workload)
        blocks = corpus.split_blocks(data, args.block_mib << 20)
        d_in = [torch.from_numpy(np.ascontiguousarray(b)).to(dev) for b in blocks]
        c0 = jam.Context(0, None)
        d_cmp, sizes = [], []
        for i, b in enumerate(blocks):
            cap = jam.ans_capacity(len(b) + jam.TRAILER)
            o = torch.empty(cap, dtype=torch.uint8, device=dev)
            m = c0.block_compress(d_in[i], len(b), o, cap)
            d_cmp.append(o[:m].clone()); sizes.append(m)
        c0.close()
        jam.shutdown()
        try:                    # first: nothing but one closed context has touched the hardware queues of this process
            caps_ = [jam.ans_capacity(len(b) + jam.TRAILER) for b in blocks]
            out["blocks_compress_call"] = blocks_compress_call(jam, torch, dev, 0, d_in, [len(b) for b in blocks], caps_, d_cmp, sizes, max(1, args.contexts))
            out["blocks_compress_call"]["process"] = "fresh child process of bench.py"
        except Exception as ex:       # noqa: BLE001
            out["blocks_compress_call"] = {"error": repr(ex)}
        out["decompress"], out["decompress_ok"] = decompress_leg(jam, torch, dev, 0, blocks, d_in, d_cmp, sizes, max(1, args.contexts), 10)
        out["decompress"]["process"] = "fresh child process of bench.py"
        try:
            out["blocks_decompress_call"] = blocks_decompress_call(jam, torch, dev, d_in[0], d_cmp[0], sizes[0])
            out["blocks_decompress_call"]["process"] = "fresh child process of bench.py"
        except Exception as ex:       # noqa: BLE001
            out["blocks_decompress_call"] = {"error": repr(ex)}
        del d_in, d_cmp
        torch.cuda.empty_cache()
        jam.shutdown()
    except Exception as ex:       # noqa: BLE001
        out["decompress"] = {"error": repr(ex)}
    try:
        out["per_block_size"] = per_block_size(jam, corpus, torch, dev, 0, max(1, args.contexts))
        out["per_block_size"]["process"] = "fresh child process of bench.py"
    except Exception as ex:       # noqa: BLE001
        out["per_block_size"] = {"error": repr(ex)}
    try:
        out["host_buffers"] = host_buffers(jam, corpus)
        out["host_buffers"]["process"] = "fresh child process of bench.py"
    except Exception as ex:       # noqa: BLE001
        out["host_buffers"] = {"error": repr(ex)}
    print("CHILD_EXTRAS " + json.dumps(out), flush=True)


def run_child_extras(args):
    """starts `bench.py --child-extras` and returns its dict (None if the child failed: the caller falls back to measuring in-process)"""
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child-extras", "--contexts", str(args.contexts), "--workload", args.workload,
                            "--block-mib", str(args.block_mib)], capture_output=True, text=True, timeout=900)
        for ln in r.stdout.splitlines():
            if ln.startswith("CHILD_EXTRAS "):
                return json.loads(ln[len("CHILD_EXTRAS "):])
    except Exception:       # noqa: BLE001
        pass
    return None


def main():
    args = parse()
    if args.child_extras:
        child_extras(args)
        return
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        sys.exit(launch_children(args))
    world = int(env_world or "1")
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with `python bench.py --gpus N` (spawns the ranks itself) "
                         f"or `python -m torch.distributed.run --nnodes=1 --nproc-per-node {args.gpus} --master-addr 127.0.0.1 bench.py --gpus {args.gpus} ...`")
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if ONE_GPU_RANKS else int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or FORCE_GATHER
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(args.master_port))
        if ONE_GPU_RANKS:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import jampack_amd as jam
    from jampack_amd import corpus, shard

    bs = args.block_mib << 20
    stream_mode = args.workload == "enwik9"          # config 4: one stream, blocks owned b mod N (strong scaling)
    if stream_mode:
        total = min(corpus.workload_bytes(args.workload), args.limit_bytes or 1 << 62)
        ranges = corpus.block_ranges(total, bs)
        mine = shard.my_blocks(len(ranges), rank, world)
        blocks, source = [], "synthetic"
        for b in mine:
            d, source = corpus.load_or_make(args.workload, limit=args.limit_bytes or None, start=ranges[b][0], count=ranges[b][1])
            blocks.append(d)
        job_bytes = total
        nblocks_job = len(ranges)
    else:
        # every rank gets its own batch of the same shape (weak scaling over independent blocks)
        data, source = corpus.load_or_make(args.workload, limit=args.limit_bytes or None, seed_offset=1000 * rank)
        blocks = corpus.split_blocks(data, bs)
        mine = list(range(len(blocks)))
        job_bytes = world * int(sum(len(b) for b in blocks))
        nblocks_job = world * len(blocks)
    batch_bytes = int(sum(len(b) for b in blocks))
    d_in = [torch.from_numpy(np.ascontiguousarray(b)).to(dev) for b in blocks]
    caps = [jam.ans_capacity(len(b) + jam.TRAILER) for b in blocks]
    d_out = [torch.empty(c, dtype=torch.uint8, device=dev) for c in caps]
    stream = torch.cuda.current_stream()
    ctx = jam.Context(local_rank, stream.cuda_stream)           # stage breakdown / profiling context (torch's stream)
    # blocks are independent (jampack.cpp:215: one Jampack instance per OpenMP thread): keep `--contexts` of them in
    # flight, each on its own context = own HBM arena + own HIP stream, driven by one host thread each
    import concurrent.futures as cf
    import queue
    import threading
    nctx = max(1, args.contexts)          # blocks in flight; with fewer blocks than contexts in a pass, consecutive passes overlap
    ctxs = [jam.Context(local_rank, None) for _ in range(nctx)]
    for c in [ctx] + ctxs:
        c.reserve(max([len(b) for b in blocks] + [1]))          # HBM arenas sized before anything is timed
    pool = cf.ThreadPoolExecutor(max_workers=nctx)
    order = sorted(range(len(blocks)), key=lambda i: -len(blocks[i]))          # largest first
    lanes = [order[k::nctx] for k in range(nctx)]
    nblk = len(blocks)
    # every context owns its output buffers (NSETS per block: the RCCL gather of step s reads while steps s+1 .. s+NSETS-1 are being written)
    NSETS = 4
    out_bufs = [[[torch.empty(caps[i], dtype=torch.uint8, device=dev) for i in range(nblk)] for _ in range(NSETS)] for _ in range(nctx)]

    sizes = [0] * nblk
    gathered = [None]
    # the largest number of blocks one rank owns: sizes the one fixed-size all_gather of (count, sizes) in shard.gather_blocks
    max_local = (nblocks_job + world - 1) // world if stream_mode else nblk
    gather_s = [0.0, 0]                  # host time this rank spent inside the gather (+ its stream sync), gathers

    def run_steps(nsteps, inputs=None, result=None, gather=True):
        """`nsteps` passes over the batch.  The (step, block) tasks go through one queue to the contexts -- a free context takes the
        next block, as the reference's OpenMP block loop over a long file does (jampack.cpp:205-224) -- so the passes follow each
        other without a host-side join; `--contexts` blocks are in flight at any time.  With more than one rank the compressed
        blocks of step s are gathered on rank 0 (the path's only exchange: one fixed-size all_gather of the sizes, then one send per
        rank of exactly its bytes, RCCL over xGMI; jampack_amd/shard.py) while step s+1 is being compressed."""
        if nsteps <= 0:
            return
        if nblk == 0:                    # more ranks than blocks: this rank still takes part in every gather
            if gather and use_dist:
                for _ in range(nsteps):
                    shard.gather_blocks([], dst=0, device=dev, max_local=max_local)
            return
        inputs = d_in if inputs is None else inputs
        q = queue.Queue()
        cond = threading.Condition()
        remaining = [nblk] * nsteps
        res = [[None] * nblk for _ in range(nsteps)]
        err = []

        def enqueue(s_):
            for i in order:
                q.put((s_, i))

        def worker(k):
            while True:
                item = q.get()
                if item is None:
                    return
                s_, i = item
                try:
                    n = ctxs[k].block_compress(inputs[i], len(blocks[i]), out_bufs[k][s_ % NSETS][i], caps[i])
                except Exception as ex:       # noqa: BLE001 -- reported by the main thread
                    err.append(ex)
                    n = -1
                with cond:
                    res[s_][i] = (k, n)
                    remaining[s_] -= 1
                    cond.notify_all()

        th = [threading.Thread(target=worker, args=(k,)) for k in range(nctx)]
        for t_ in th:
            t_.start()
        gather = gather and use_dist
        depth = NSETS if gather else nsteps        # steps in flight, the one whose gather is pending included
        for s_ in range(min(depth, nsteps)):
            enqueue(s_)
        for s_ in range(nsteps):
            with cond:
                cond.wait_for(lambda: remaining[s_] == 0)
            if gather:
                tg0 = time.perf_counter()
                gathered[0] = shard.gather_blocks([out_bufs[res[s_][i][0]][s_ % NSETS][i][: max(res[s_][i][1], 0)] for i in range(nblk)], dst=0, device=dev,
                                                  max_local=max_local)
                torch.cuda.current_stream().synchronize()      # the gather has read this buffer set before step s+NSETS may write it
                gather_s[0] += time.perf_counter() - tg0
                gather_s[1] += 1
            if s_ + depth < nsteps:
                enqueue(s_ + depth)
        for _ in th:
            q.put(None)
        for t_ in th:
            t_.join()
        if err:
            raise err[0]
        last = res[nsteps - 1]
        for i in range(nblk):
            if result is not None:
                result[i] = last[i][1]
            else:
                sizes[i] = last[i][1]
                d_out[i] = out_bufs[last[i][0]][(nsteps - 1) % NSETS][i]

    def lane_work(k):
        for i in lanes[k]:
            sizes[i] = ctxs[k].block_compress(d_in[i], len(blocks[i]), d_out[i], caps[i])

    def compress_step():               # one pass with a host-side join at its end (round 1's timed loop; kept as an extra)
        for f in [pool.submit(lane_work, k) for k in range(nctx)]:
            f.result()

    def sync_all():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # Every context compresses every block of the batch once before anything is counted (round 6).  The W warm-up steps go through the
    # same work queue as the timed ones: with ten contexts and two blocks per step, W = 5 steps are ten tasks, and a context that happens
    # to take none of them would make its first call -- stream and event creation, the encoder's second arena layout, pinned mailboxes --
    # inside the timed region.  Inputs stay resident; the timed region is exactly K steps, behind exactly W warm-up steps.
    def prime(k):
        for i in order:
            ctxs[k].block_compress(d_in[i], len(blocks[i]), out_bufs[k][0][i], caps[i])
    if nblk:
        pth = [threading.Thread(target=prime, args=(k,)) for k in range(nctx)]
        for t_ in pth:
            t_.start()
        for t_ in pth:
            t_.join()
        torch.cuda.synchronize()
    run_steps(args.warmup)
    sync_all()
    gather_s[0], gather_s[1] = 0.0, 0
    t0 = time.perf_counter()
    run_steps(args.steps)
    sync_all()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ms_per_step = dt / args.steps * 1e3
    value = job_bytes / 1e6 / (dt / args.steps)

    extra = {}
    if rank == 0 and world == 1 and not args.no_extras and blocks:
        # the workload variants first: the same loop, the same contexts, right behind the timed steps (before the extras that create contexts
        # and streams of their own: a later leg of this process measures 5-10 % lower, see child_extras)
        if args.workload == "enwik8" and not args.limit_bytes:
            # the same step on the phrase-book text (deeper repeats, ratio ~10 %): round 1's headline corpus, kept for comparison
            pdata, _ = corpus.load_or_make("enwik8-phrase")
            pblocks = corpus.split_blocks(pdata, bs)
            p_in = [torch.from_numpy(np.ascontiguousarray(b)).to(dev) for b in pblocks]
            psz = [0] * len(pblocks)

            run_steps(3, p_in, psz, gather=False)
            torch.cuda.synchronize()
            tp0 = time.perf_counter()
            run_steps(10, p_in, psz, gather=False)
            torch.cuda.synchronize()
            tp = (time.perf_counter() - tp0) / 10
            extra["phrase_book_variant"] = {"value": round(batch_bytes / 1e6 / tp, 1), "unit": "MB/s", "ms_per_step": round(tp * 1e3, 3), "steps": 10,
                                            "compressed_ratio": round(sum(psz) / batch_bytes, 4),
                                            "workload": "same shape, text with a 200 000-phrase book (round 1's corpus)"}
            del p_in
            # ... and on a text with enwik8's BYTE ALPHABET: the headline text has 28 distinct bytes, which lets round 0 of
            # the suffix sort key on 11 bytes; real enwik8 has 205 (capitals, digits, punctuation, markup, UTF-8 pairs: order-0 entropy
            # 5.1 bits) and gets the 7-byte keys of any block above 128 byte values.  Same shape, same word model, wide spelling.
            wdata, _ = corpus.load_or_make("enwik8-wide")
            wblocks = corpus.split_blocks(wdata, bs)
            w_in = [torch.from_numpy(np.ascontiguousarray(b)).to(dev) for b in wblocks]
            wsz = [0] * len(wblocks)
            hist = np.bincount(wdata, minlength=256).astype(np.float64)
            pr = hist[hist > 0] / hist.sum()
            run_steps(3, w_in, wsz, gather=False)
            torch.cuda.synchronize()
            tw0 = time.perf_counter()
            run_steps(12, w_in, wsz, gather=False)
            torch.cuda.synchronize()
            tw = (time.perf_counter() - tw0) / 12
            extra["wide_alphabet_variant"] = {"value": round(batch_bytes / 1e6 / tw, 1), "unit": "MB/s", "ms_per_step": round(tw * 1e3, 3), "steps": 12,
                                              "compressed_ratio": round(sum(wsz) / batch_bytes, 4), "alphabet": int((hist > 0).sum()),
                                              "order0_entropy_bits": round(float(-(pr * np.log2(pr)).sum()), 3),
                                              "workload": "same shape and word model over an enwik8-like byte alphabet (capitals, digits, punctuation, markup, "
                                                          "UTF-8 pairs): a fixed-width code holds 7 of its bytes per sort key (11 of the headline text's), the variable-length code about 10 (12), the order-2 context codes 13-14 (15)"}
            del w_in, wdata
            run_steps(1, gather=False)          # the variants wrote the loop's output buffers: the workload's own blocks again (sizes, d_out)
            torch.cuda.synchronize()
    if use_dist:
        backend = dist.get_backend()
        extra["collective"] = {"backend": ("rccl (torch.distributed nccl)" if backend == "nccl" else backend), "ranks": dist.get_world_size(),
                               "gathers_timed": gather_s[1],
                               "gather_ms_per_step": round(gather_s[0] / max(gather_s[1], 1) * 1e3, 3),
                               "note": "host time rank 0 spends in one gather of a step's compressed blocks (fixed-size all_gather of the sizes + one exact-size "
                                       "send per rank + stream sync); it overlaps the compression of the following steps"}
    if rank == 0 and gathered[0] is not None:
        # rank 0 holds every rank's compressed blocks; in stream mode they are put back into file order (what the in-order
        # CompWriteBlock loop writes, jampack.cpp:220-224) and rank 0's own blocks are checked against their sources
        per_rank = gathered[0]
        ok = len(per_rank) == world and all(bool(torch.equal(per_rank[0][k], d_out[k][: sizes[k]])) for k in range(len(blocks)))
        if stream_mode:
            ordered = shard.assemble_in_block_order(per_rank, nblocks_job)
            ok = ok and len(ordered) == nblocks_job and all(bool(torch.equal(ordered[b], d_out[k][: sizes[k]])) for k, b in enumerate(mine))
            extra["gathered_payload_bytes"] = int(sum(t.numel() for t in ordered))
        extra["gather_ok"] = bool(ok)

    def roofline_extra():
        """`roofline` (+ `roofline_kernels`) of the JSON line, measured on this rank"""
        extra_ = {}
        # roofline: HIP events around every kernel on its launch stream (the library's profiler), (a) over passes of the timed loop
        # itself -- the same contexts, the same blocks in flight, so the launch shape and the contention are those of `value` -- and
        # (b) over one pass with one block at a time.  Both run BEFORE the extras that create more contexts: streams are dealt onto
        # the hardware queues round robin at creation, and a chain kernel whose stream shares a queue with another stream is timed
        # from its event, i.e. including its wait in the queue.
        def prof_rows(tabs):
            acc = {}
            for tab in tabs:
                for r in tab:
                    a_ = acc.setdefault(r["name"], {"ms": 0.0, "launches": 0, "units": 0})
                    a_["ms"] += r["ms"]; a_["launches"] += r["launches"]; a_["units"] += r["units"]
            rows = []
            for name, r in acc.items():
                bpu = ALG_BYTES_PER_UNIT.get(name)
                if not bpu or not r["units"] or r["ms"] <= 0:
                    continue
                ach = bpu[0] * r["units"] / 1e9 / (r["ms"] / 1e3)
                rows.append({"kernel": name, "ms_total": round(r["ms"], 3), "launches": r["launches"], "avg_launch_us": round(r["ms"] * 1e3 / r["launches"], 2),
                             "alg_bytes_per_unit": bpu[0], "unit_is": bpu[1], "units": r["units"], "achieved": round(ach, 2), "frac": round(ach / 8000.0, 5),
                             "limited_by": bpu[2]})
            rows.sort(key=lambda r: -r["ms_total"])
            return rows

        for c_ in ctxs:
            c_.profile_enable(2)
        run_steps(4, gather=False)
        torch.cuda.synchronize()
        rows = prof_rows([c_.profile_table() for c_ in ctxs])
        for c_ in ctxs:
            c_.profile_enable(0)
        # the chain against the yardstick of what bounds it: a chain is ONE wave on a serial recurrence -- its HBM
        # fraction says nothing; what it can reach is the issue rate of a lone wave.  Cycles per step of the slowest chunk of each context's
        # last block in the loop (s_memtime stamps inside k_rans_lanes, jpk_stats.enc_chain_*), against the step's dependent path: 11 of
        # its 12 vector instructions wait for the one before them, a lone wave issues a dependent instruction every 5.2 cycles
        # (tools/issuetest.hip, profiles/r05_issuetest.txt), + 1/16 of the ~25 instructions between two batches of sixteen steps.
        cps = []
        for c_ in ctxs:
            st_ = c_.stats()
            if st_.enc_chain_steps > 0 and st_.enc_chain_cycles > 0:
                cps.append(st_.enc_chain_cycles / st_.enc_chain_steps)
        floor_cps = 11 * 5.2
        issue = None
        if cps:
            mean_cps = sum(cps) / len(cps)
            issue = {"cycles_per_step": round(mean_cps, 1), "lone_wave_floor_cycles_per_step": round(floor_cps, 1), "frac": round(floor_cps / mean_cps, 3),
                     "dependent_instructions_per_step": 11, "lone_wave_cycles_per_dependent_instruction": 5.2,
                     "note": "issue-rate fraction of the chain in the timed loop: floor / measured cycles per step (1.0 = a lone wave's issue limit; below: the "
                             "chain shares its SIMD with the other blocks' kernels)"}
        ctx.profile_enable(2)
        for i, b in enumerate(blocks):
            ctx.block_compress(d_in[i], len(b), d_out[i], caps[i])
        rows1 = prof_rows([ctx.profile_table()])
        ctx.profile_enable(0)
        if rows:
            # Fixed keys: `roofline` = the rANS chain class (one launch per block in the loop, the serial floor of the
            # encode stage), `roofline_wide` = the radix scatter (the widest machine-filling class) -- not max() over totals that tie
            # within a few per cent from run to run (round 4's driver line named the scatter, its profiles the chain: a 10x swing of
            # `frac` from a tie-break).  `largest_total_time_class` says which class led this time.
            d0 = next((r for r in rows if r["kernel"] == "k_rans_lanes"), rows[0])
            traffic, pmc_file = pmc_traffic(d0["kernel"], 4, d0["launches"])
            extra_["roofline"] = {"bound": "hbm" if d0["limited_by"].startswith("hbm") else d0["limited_by"], "achieved": d0["achieved"], "peak": 8000.0, "unit": "GB/s",
                                 "frac": d0["frac"], "traffic": traffic, "traffic_source": f"committed profile profiles/{pmc_file} (rocprofv3 --pmc passes of this command; not measured in this run)" if pmc_file else None,
                                 "issue": issue,
                                 "kernel": d0["kernel"], "limited_by": d0["limited_by"], "avg_launch_us": d0["avg_launch_us"], "launches": d0["launches"],
                                 "alg_bytes_per_launch": round(d0["alg_bytes_per_unit"] * d0["units"] / d0["launches"]),
                                 "largest_total_time_class": rows[0]["kernel"],
                                 "whole_path": {"bytes_per_byte": 14.2, "achieved": round(14.2 * job_bytes / 1e9 / (ms_per_step / 1e3), 2),
                                                "frac": round(14.2 * job_bytes / 1e9 / (ms_per_step / 1e3) / 8000.0, 5),
                                                "note": "SURVEY 8d: 14.2 algorithmic bytes per block byte over the timed loop's ms_per_step"},
                                 "note": f"the rANS chain class (fixed key), over 4 passes of the timed loop ({nctx} blocks in flight: one chain launch per block, stretched by the other blocks' kernels); achieved = algorithmic bytes per launch / mean launch time (HIP events on the launch stream); traffic = PMC bytes per launch of the same command from profiles/{pmc_file}; peak = HBM spec; limited_by says what the class is really bound by (DESIGN.md section 4): a chain is 65 waves on a serial recurrence, its HBM fraction says nothing about the machine -- roofline_wide is the machine-filling class"}
            extra_["roofline_kernels"] = rows[:8]
            # the same for the radix scatter, the dominant class among those whose grids FILL the machine (the chain above is 65 waves
            # on 1024 SIMDs: its launches are long, its share of the chip is 6 %)
            wide = next((r for r in rows if r["kernel"] == "k_rs_scatter/k_os_scatter"), None) or next((r for r in rows if r["limited_by"] != "issue"), None)
            if wide:
                wt, wf = pmc_traffic(wide["kernel"], 4, wide["launches"])
                extra_["roofline_wide"] = {"kernel": wide["kernel"], "bound": "hbm" if wide["limited_by"].startswith("hbm") else wide["limited_by"],
                                          "achieved": wide["achieved"], "peak": 8000.0, "unit": "GB/s", "frac": wide["frac"], "traffic": wt,
                                          "traffic_source": f"committed profile profiles/{wf} (not measured in this run)" if wf else None,
                                          "avg_launch_us": wide["avg_launch_us"], "launches": wide["launches"],
                                          "alg_bytes_per_launch": round(wide["alg_bytes_per_unit"] * wide["units"] / wide["launches"]),
                                          "note": "the radix scatter class (fixed key): the widest machine-filling class, same 4 passes of the timed loop, same accounting"}
            one = next((r for r in rows1 if r["kernel"] == d0["kernel"]), None)
            if one:
                extra_["roofline"]["one_block_at_a_time"] = {"avg_launch_us": one["avg_launch_us"], "launches": one["launches"], "achieved": one["achieved"], "frac": one["frac"],
                                                            "traffic": pmc_traffic(d0["kernel"], 1, one["launches"])[0],
                                                            "alg_bytes_per_launch": round(one["alg_bytes_per_unit"] * one["units"] / one["launches"]),
                                                            "note": "a block alone cuts its chains into four graded launches"}
            onew = next((r for r in rows1 if wide and r["kernel"] == wide["kernel"]), None)
            if onew:
                extra_["roofline_wide"]["one_block_at_a_time"] = {"avg_launch_us": onew["avg_launch_us"], "launches": onew["launches"], "achieved": onew["achieved"],
                                                                 "frac": onew["frac"]}
        return extra_

    if rank == 0 and world > 1 and not args.no_extras and blocks:
        extra.update(roofline_extra())          # the other ranks wait at the final barrier meanwhile
    # ---- un-timed extras on rank 0 of a 1-GPU run: stage breakdown, decompress leg, parity flags, roofline, CPU baseline ----
    if rank == 0 and world == 1 and not args.no_extras and blocks:
        reps = max(3, min(args.steps, 10))
        ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731  (same stream as the library's launches)
        names = ("forward_bwt", "ans_encode", "ans_decode", "inverse_bwt")
        stage_ms = dict.fromkeys(names, 0.0)
        comp_sizes = []
        ok = True
        for i, b in enumerate(blocks):
            n = len(b)
            d_bwt = torch.empty(n + jam.TRAILER, dtype=torch.uint8, device=dev)
            d_enc = torch.empty(caps[i], dtype=torch.uint8, device=dev)
            d_dec = torch.empty(n + jam.TRAILER, dtype=torch.uint8, device=dev)
            d_back = torch.empty(max(n, 1), dtype=torch.uint8, device=dev)
            acc = dict.fromkeys(names, 0.0)
            for rep in range(reps + 1):                  # rep 0 warms every path of this context (arena growth, code load)
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
                if rep:
                    for k, name in enumerate(names):
                        acc[name] += e[k].elapsed_time(e[k + 1])
            for name in names:
                stage_ms[name] += acc[name] / reps
            comp_sizes.append(clen)
            ok = ok and bl == n and bool(torch.equal(d_back[:n], d_in[i])) and clen == sizes[i] and bool(torch.equal(d_enc[:clen], d_out[i][:clen]))
            if i == 0:
                st = ctx.stats()
                nr = int(st.sa_rounds)
                dep = int(st.sa_key_depth)
                extra["sa_rounds"] = {"block_bytes": n, "rounds": nr, "key_depth_bytes": dep, "key_order": int(st.sa_key_order), "alphabet": int(len(np.unique(blocks[i]))),
                                      "active_suffixes": [int(x) for x in st.sa_round_active[:nr]],
                                      "in_large_groups": [int(x) for x in st.sa_round_large[:nr]],
                                      "pair_rounds": [r for r in range(min(nr, 64)) if (int(st.sa_pair_rounds) >> r) & 1],
                                      "note": "round 0 = radix sort on the first key_depth_bytes symbols of every suffix on average (an order-preserving "
                                              "prefix code of the block's byte values, every symbol coded behind the one or two bytes in front of it where a "
                                              "sample of the block says that pays; as many symbols as fit 56 bits -- the fixed-width code of round 4 held 11 "
                                              "bytes of this text, 7 of any block above 128 byte values); round r >= 1 sorts the "
                                              "still unresolved suffixes, every group at its own depth, by the rank of the suffix that many symbols "
                                              "further; pair_rounds = rounds that resolved long repeats by induction from their successors instead"}
            del d_bwt, d_enc, d_dec, d_back
        mb = batch_bytes / 1e6
        extra["stages_ms"] = {k: round(v, 3) for k, v in stage_ms.items()}
        extra["stages_MBps"] = {k: round(mb / (v / 1e3), 1) for k, v in stage_ms.items() if v > 0}
        extra["stages_reps"] = reps
        extra.update(roofline_extra())
        # the library's own blocks-in-flight loop through ONE call: measured in the fresh child process (child_extras) when there is one -- the
        # way a caller's process sees it; a context created HERE, behind the timed loop's eleven, shares hardware queues with theirs and has
        # measured anything from 3.8 to 6.0 GB/s on boxes whose child process gives 6.0-6.3 -- and here otherwise
        if args.no_block_sizes or args.limit_bytes:
            extra["blocks_compress_call"] = blocks_compress_call(jam, torch, dev, local_rank, d_in, [len(b_) for b_ in blocks], caps, d_out, sizes, nctx)
            extra["blocks_compress_call"]["process"] = "the bench process, behind the timed loop"
        # decompress leg (rANS decode -> inverse BWT, jampack.cpp:49-50) over the same batch, blocks in flight like compress: measured in
        # the fresh child process (child_extras) when there is one, here otherwise
        d_cmp = [d_out[i][: sizes[i]].clone() for i in range(len(blocks))]
        ce = None
        if not args.no_block_sizes and not args.limit_bytes:
            ce = run_child_extras(args)
        if ce is not None and isinstance(ce.get("blocks_compress_call"), dict) and "value" in ce["blocks_compress_call"]:
            extra["blocks_compress_call"] = ce["blocks_compress_call"]
        elif "blocks_compress_call" not in extra:          # (the child failed: here after all)
            extra["blocks_compress_call"] = blocks_compress_call(jam, torch, dev, local_rank, d_in, [len(b_) for b_ in blocks], caps, d_out, sizes, nctx)
            extra["blocks_compress_call"]["process"] = "the bench process, behind the timed loop"
        if ce is not None and isinstance(ce.get("decompress"), dict) and "value" in ce["decompress"]:
            dleg, dok_ = ce["decompress"], bool(ce.get("decompress_ok", False))
        else:
            dleg, dok_ = decompress_leg(jam, torch, dev, local_rank, blocks, d_in, d_cmp, sizes, nctx, reps)
        ok = ok and dok_
        dleg["one_block_at_a_time_MBps"] = round(mb / ((stage_ms["ans_decode"] + stage_ms["inverse_bwt"]) / 1e3), 1)
        dleg["inverse_bwt_MBps"] = round(mb / (stage_ms["inverse_bwt"] / 1e3), 1)
        extra["decompress"] = dleg
        extra["round_trip_ok"] = ok
        del d_cmp
        # BASELINE config 3's "120-way parallel LF-map": the reference's own GPU kernel shape (CUDAInverse<<<40,3>>>, bwt.cpp:8-19) as
        # a measured comparator beside the list-ranking inverse, on block 0
        try:
            n0 = len(blocks[0])
            d_b0 = torch.empty(n0 + jam.TRAILER, dtype=torch.uint8, device=dev)
            d_t0 = torch.empty(max(n0, 1), dtype=torch.uint8, device=dev)
            ctx.bwt_forward(d_in[0], n0, d_b0, n0 + jam.TRAILER)
            m0, chase_ms = ctx.bwt_inverse_chains120(d_b0, n0 + jam.TRAILER, d_t0, n0)
            e0, e1 = ev(), ev()
            e0.record(stream)
            ctx.bwt_inverse(d_b0, n0 + jam.TRAILER, d_t0, n0)
            e1.record(stream)
            torch.cuda.synchronize()
            extra["inverse_bwt_120_chains"] = {"block_bytes": n0, "chase_kernel_ms": round(chase_ms, 3), "MBps": round(n0 / 1e6 / (chase_ms / 1e3), 1),
                                               "list_ranking_inverse_ms": round(e0.elapsed_time(e1), 3), "same_bytes": bool(m0 == n0),
                                               "note": "120 threads, p = Map[p-1] (the reference's CUDAInverse shape) on the same Map; comparator only"}
            del d_b0, d_t0
        except Exception as ex:
            extra["inverse_bwt_120_chains"] = {"error": repr(ex)}
        # round 1's timed loop for comparison: every pass ends with a host-side join of the contexts
        compress_step()
        torch.cuda.synchronize()
        tb0 = time.perf_counter()
        for _ in range(reps):
            compress_step()
        torch.cuda.synchronize()
        tb = (time.perf_counter() - tb0) / reps
        extra["join_per_step"] = {"value": round(mb / tb, 1), "unit": "MB/s", "ms_per_step": round(tb * 1e3, 3), "steps": reps,
                                  "note": "the same passes with all contexts joined after every pass (the timed loop of round 1)"}
        extra["compressed_bytes"] = int(sum(comp_sizes))
        extra["compressed_ratio"] = round(sum(comp_sizes) / batch_bytes, 4)
        extra["workspace_bytes"] = int(ctx.stats().workspace_bytes)
        # algorithmic HBM traffic of the stages: fwd BWT 10 B/B, ANS 4+c B/B, inverse BWT 12 B/B
        c = sum(comp_sizes) / batch_bytes
        sg = {k: STAGE_ALG[k](c) * batch_bytes / 1e9 / (stage_ms[k] / 1e3) for k in names if stage_ms[k] > 0}
        extra["stages_alg_GBps"] = {k: {"achieved": round(v, 2), "frac": round(v / 8000.0, 5)} for k, v in sg.items()}
        comp_ms = stage_ms["forward_bwt"] + stage_ms["ans_encode"]
        extra["compress_alg"] = {"bytes_per_byte": round(14.0 + c, 2), "achieved_GBps": round((14.0 + c) * batch_bytes / 1e9 / (ms_per_step / 1e3), 2),
                                 "frac": round((14.0 + c) * batch_bytes / 1e9 / (ms_per_step / 1e3) / 8000.0, 5), "one_block_at_a_time_ms": round(comp_ms, 3)}
        # per-kernel HIP-event timing (events recorded by the library on the launch stream) of one more compress pass
        # The extras below bring contexts of their own.  The loop's contexts (each with its stream and up to three encoder group
        # streams) are closed first: HIP deals streams onto 32 hardware queues, a stream beyond that shares a queue, and a 12 ms
        # chain kernel then blocks whatever sits behind it -- with the loop's twenty streams still alive the 8 MiB leg of
        # per_block_size measured 2.8 GB/s where a fresh process measures 3.6 (tools/small_blocks.py).
        pool.shutdown()
        for c_ in ctxs:
            c_.close()
        torch.cuda.empty_cache()
        if not args.no_block_sizes and not args.limit_bytes:
            if ce is not None:                   # (the fresh child process has measured them: see child_extras())
                extra.update({k: v for k, v in ce.items() if k in ("per_block_size", "host_buffers", "blocks_decompress_call")})
            else:
                try:
                    extra["per_block_size"] = per_block_size(jam, corpus, torch, dev, local_rank, nctx)
                except Exception as ex:       # noqa: BLE001 -- an extra must never take the headline down
                    extra["per_block_size"] = {"error": repr(ex)}
                try:
                    extra["host_buffers"] = host_buffers(jam, corpus)
                except Exception as ex:       # noqa: BLE001
                    extra["host_buffers"] = {"error": repr(ex)}
        if not args.no_cpu_baseline:
            try:
                cb, ref_enc, n = cpu_baseline(blocks, args.cpu_sample_mib)
                # bit-exact vs the CPU reference on the same sample block
                d_s = d_in[0][:n].contiguous()
                d_o = torch.empty(jam.ans_capacity(n + jam.TRAILER), dtype=torch.uint8, device=dev)
                m = ctx.block_compress(d_s, n, d_o, d_o.numel())
                cb["bit_exact_vs_cpu"] = bool(m == len(ref_enc) and np.array_equal(d_o[:m].cpu().numpy(), ref_enc))
                extra["cpu_baseline"] = cb
            except Exception as ex:  # the baseline must never take the GPU line down
                extra["cpu_baseline"] = {"value": None, "unit": "MB/s", "cores": 0, "kind": "error", "sample": repr(ex)}

    if rank == 0:
        # the strings say what was timed: a gather only when one ran, under the name of the backend that carried it
        coll = ""
        if use_dist:
            coll = ("RCCL" if dist.get_backend() == "nccl" else dist.get_backend()) + " gather of the compressed blocks on rank 0"
        if stream_mode:
            wl = (f"{args.workload}-like: ONE {job_bytes} B stream as {args.block_mib} MiB blocks ({nblocks_job} blocks), block b on GPU b mod {world}, "
                  "forward BWT + rANS encode, inputs resident in HBM" + (f", {coll} in block order" if use_dist else ", no gather (one rank)"))
            par = f"{nblocks_job} blocks sharded b mod {world} over {world} GPU(s)" + (f", {coll}" if use_dist else "")
        else:
            wl = (f"{args.workload}-like {batch_bytes} B per GPU as {args.block_mib} MiB blocks ({len(blocks)} blocks), forward BWT + rANS encode, inputs resident in HBM")
            par = (f"one batch per GPU on {world} GPUs, {coll}" if world > 1 else ("1 GPU" + (f", {coll} (forced, world 1)" if use_dist else "")))
        if ONE_GPU_RANKS and world > 1:
            par += " [test hook JPK_BENCH_ONE_GPU: all ranks share cuda:0 -- exercises the N > 1 code path, not a scaling number]"
        line = {
            "metric": "MB/s compress (forward BWT + rANS encode) on 64 MiB blocks; bit-exact vs CPU ref",
            "value": round(value, 2), "unit": "MB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "strong" if stream_mode else "weak", "vs_baseline": None,
            "dtype": "u8/int32", "data": "synthetic" if source == "synthetic" else source,
            "config": {"workload": wl + ("" if not args.limit_bytes else " [TRUNCATED: not a valid headline]"),
                       "block_bytes": [len(b) for b in blocks] if not stream_mode else [r[1] for r in ranges],
                       "parallelism": par + f", {nctx} blocks in flight per GPU, passes enqueued back to back through a work queue (no host join between steps)"},
        }
        line.update(extra)
        # RCCL prints its version banner through C stdio, which sits in a buffer until exit when stdout is a pipe: push it out
        # first so that the JSON line is the LAST line of stdout
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:       # noqa: BLE001
            pass
        print(json.dumps(line), flush=True)
    pool.shutdown()
    for c in ctxs:
        c.close()
    ctx.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
