#!/usr/bin/env python3
"""Interference matrix: which resource do the blocks in flight compete for?

rocprofv3 counter collection serialises the dispatches (profiles/r04_loop_counters.txt: 1.1 GB/s under --pmc against 3.9 free), so
counters cannot show contention.  This does it by experiment: the forward BWT alone / the rANS encode alone / the whole compress,
N contexts in flight (tools/stage_scaling.py's loop), beside a persistent hog (tools/hog.hip) that occupies `wgs` workgroups of 256
threads per CU with ONE kind of work.  The slowdown per hog kind says what the stage is short of.

   hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/hog.hip -o tools/_bin/libhog.so
   python tools/interfere.py [contexts=4] [wgs_per_cu=2]"""
import ctypes, os, sys, threading, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")      # before the HIP runtime starts: the hog needs a hardware queue of its own
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import jampack_amd as jam

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4
WGS = int(sys.argv[2]) if len(sys.argv) > 2 else 2
hog = ctypes.CDLL(os.path.join(ROOT, "tools", "_bin", "libhog.so"))
hog.hog_stop.restype = ctypes.c_double
cus = hog.hog_init()
assert cus > 0, cus
n = 64 << 20
t = jam.corpus.make("text_survey", n, 8)
dev = torch.device("cuda", 0)
d_in = torch.from_numpy(t).to(dev)
cap = jam.ans_capacity(n + 480)
ctx0 = jam.Context(0, None)
d_bwt = torch.empty(n + 480, dtype=torch.uint8, device=dev)
ctx0.bwt_forward(d_in, n, d_bwt, n + 480)
KINDS = [(-1, "none"), (4, "sleep (wave slots only)"), (0, "valu"), (1, "lds"), (2, "stream 16 B/lane"), (3, "gather 4 B random")]
UNIT = {0: ("G mad/s per lane-sum", 1e9), 1: ("G lds acc/s", 1e9), 2: ("GB/s", 1e9), 3: ("G gathers/s", 1e9), 4: ("M sleeps/s", 1e6)}
print(f"# {cus} CUs; hog = {WGS} workgroups of 256 threads per CU ({WGS} waves per SIMD); {N} contexts in flight; 64 MiB text block")


def run(mode):
    ctxs = [jam.Context(0, None) for _ in range(N)]
    outs = [torch.empty(cap, dtype=torch.uint8, device=dev) for _ in range(N)]
    reps = 5

    def work(k):
        for _ in range(reps):
            if mode == "fwd":
                ctxs[k].bwt_forward(d_in, n, outs[k], n + 480)
            elif mode == "enc":
                ctxs[k].ans_encode(d_bwt, n + 480, outs[k], cap)
            else:
                ctxs[k].block_compress(d_in, n, outs[k], cap)

    for c in range(N):
        ctxs[c].bwt_forward(d_in, n, outs[c], n + 480)
        ctxs[c].ans_encode(d_bwt, n + 480, outs[c], cap)
    base = None
    for kind, name in KINDS:
        torch.cuda.synchronize()
        if kind >= 0:
            assert hog.hog_start(kind, WGS) == 0
            time.sleep(0.02)
        t0 = time.perf_counter()
        th = [threading.Thread(target=work, args=(k,)) for k in range(N)]
        [x.start() for x in th]; [x.join() for x in th]
        dt = time.perf_counter() - t0
        w = hog.hog_stop() if kind >= 0 else 0.0
        torch.cuda.synchronize()
        per = dt / (N * reps) * 1e3
        if base is None:
            base = per
        rate = ""
        if kind >= 0:
            u, s = UNIT[kind]
            mult = 1.0 if kind != 2 else 1.0
            rate = f"   hog did {w * mult / dt / s:9.1f} {u}"
        print(f"{mode:5s} beside {name:26s}: {per:7.2f} ms per block  x{per / base:5.2f}{rate}", flush=True)
    for c in ctxs:
        c.close()


# the hogs alone (what the chip gives them with nothing else running), 0.3 s each
for kind, name in KINDS[1:]:
    hog.hog_start(kind, WGS); t0 = time.perf_counter(); time.sleep(0.3); w = hog.hog_stop(); dt = time.perf_counter() - t0
    u, s = UNIT[kind]
    print(f"hog alone {name:26s}: {w / dt / s:9.1f} {u}", flush=True)
for mode in ("fwd", "enc", "both"):
    run(mode)
