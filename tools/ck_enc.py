import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')
import numpy as np, torch
import jampack_amd as jam
n = 64 << 20
t = jam.corpus.make("text", n, 8)
dev = torch.device("cuda", 0); st = torch.cuda.current_stream()
ctx = jam.Context(0, st.cuda_stream)
d_in = torch.from_numpy(t).to(dev); cap = jam.ans_capacity(n + 480)
d_bwt = torch.empty(n + 480, dtype=torch.uint8, device=dev); d_enc = torch.empty(cap, dtype=torch.uint8, device=dev)
ctx.bwt_forward(d_in, n, d_bwt, n + 480)
ctx.ans_encode(d_bwt, n + 480, d_enc, cap)
for mode in ("wall", "prof"):
    if mode == "prof": ctx.profile_enable(2)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(3): ctx.ans_encode(d_bwt, n + 480, d_enc, cap)
    e1.record(st); torch.cuda.synchronize()
    print(mode, "wall ms/call", (time.perf_counter() - t0) / 3 * 1e3, "event ms/call", e0.elapsed_time(e1) / 3)
    if mode == "prof":
        for r in sorted(ctx.profile_table(), key=lambda r: -r["ms"])[:4]: print("   ", r["name"], r["ms"] / 3)
