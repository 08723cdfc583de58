#!/usr/bin/env python3
"""focused checks while the run-aware sort and the grouped compress path are being brought up (GPU box)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
import numpy as np, torch
import jampack_amd as jam
from oracle.pyoracle import Oracle

dev = torch.device("cuda", 0)
st = torch.cuda.current_stream()
ctx = jam.Context(0, st.cuda_stream)
o = Oracle()
what = sys.argv[1:] or ["zero", "stages", "group"]

if "zero" in what:
    for n in (120_000, 4 << 20, 64 << 20):
        t = np.zeros(n, np.uint8)
        d_in = torch.from_numpy(t).to(dev)
        d_bwt = torch.empty(n + 480, dtype=torch.uint8, device=dev)
        t0 = time.perf_counter(); ctx.bwt_forward(d_in, n, d_bwt, n + 480); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        got = d_bwt.cpu().numpy()
        nlen = n - n % 120; step = nlen // 120
        exp_tr = (nlen - np.arange(120) * step).astype("<i4")
        tr = got[n:].view("<i4")
        s = ctx.stats()
        print(f"zero n={n}: image zero={not got[:n].any()} trailer ok={np.array_equal(tr, exp_tr)} rounds={s.sa_rounds} {dt * 1e3:.1f} ms", flush=True)
        if not np.array_equal(tr, exp_tr):
            print("   trailer", tr[:6], "expected", exp_tr[:6])

if "stages" in what:
    for kind, n in (("zero", 64 << 20), ("runs", 8 << 20), ("dna", 70000), ("dna", 8 << 20), ("two", 300000), ("repeat4k", 300000), ("silesia", 16 << 20)):
        t = jam.corpus.make(kind, n, 21)
        d_in = torch.from_numpy(t).to(dev); cap = jam.ans_capacity(n + 480)
        d_bwt = torch.empty(n + 480, dtype=torch.uint8, device=dev); d_enc = torch.empty(cap, dtype=torch.uint8, device=dev)
        d_dec = torch.empty(n + 480, dtype=torch.uint8, device=dev); d_back = torch.empty(n, dtype=torch.uint8, device=dev)
        line = f"{kind:9s} n={n}:"
        try:
            ctx.bwt_forward(d_in, n, d_bwt, n + 480); line += f" fwd rounds={ctx.stats().sa_rounds}"
            img = d_bwt.clone()
            if n <= 1 << 20:
                exp = o.bwt_forward(t)
                line += f" fwd==oracle {np.array_equal(d_bwt.cpu().numpy(), exp)}"
            cl = ctx.ans_encode(d_bwt, n + 480, d_enc, cap); line += f" enc {cl}"
            dl = ctx.ans_decode(d_enc, cl, d_dec, n + 480); line += f" dec==img {bool(torch.equal(d_dec, img))}"
            bl = ctx.bwt_inverse(img, n + 480, d_back, n); line += f" inv==in {bool(torch.equal(d_back, d_in))}"
        except Exception as e:  # noqa: BLE001
            line += f" EXC {e}"
        print(line, flush=True)

if "group" in what:
    MiB = 1 << 20
    data = jam.corpus.make("text_survey", 16 * MiB, 12)
    for bs, nb in ((MiB, 4), (MiB, 16), (2 * MiB + 77, 5), (300_001, 7)):
        blocks = [np.ascontiguousarray(data[k * bs: (k + 1) * bs]) for k in range(nb)]
        d_ins = [torch.from_numpy(b).to(dev) for b in blocks]
        cap = jam.ans_capacity(bs + 480)
        outs = [torch.empty(cap, dtype=torch.uint8, device=dev) for _ in range(nb)]
        n, stt = ctx.blocks_compress(d_ins, [bs] * nb, outs, [cap] * nb, 2)
        bad = []
        for k in range(nb):
            one = torch.empty(cap, dtype=torch.uint8, device=dev)
            m = ctx.block_compress(d_ins[k], bs, one, cap)
            if m != n[k] or not torch.equal(one[:m], outs[k][:m]):
                a, b = one[:m].cpu().numpy(), outs[k][: n[k]].cpu().numpy()
                mm = min(len(a), len(b)); d = np.nonzero(a[:mm] != b[:mm])[0]
                bad.append((k, m, n[k], d[:3].tolist(), len(d)))
        print(f"group bs={bs} nb={nb}: status {stt} mismatches {bad}", flush=True)
    # the image alone: group forward BWT through the fused call is not exposed; compare a 2-block group against the oracle stream
    blocks = [np.ascontiguousarray(data[:200_000]), np.ascontiguousarray(data[200_000:500_000])]
    d_ins = [torch.from_numpy(b).to(dev) for b in blocks]
    caps = [jam.ans_capacity(len(b) + 480) for b in blocks]
    outs = [torch.empty(c, dtype=torch.uint8, device=dev) for c in caps]
    n, stt = ctx.blocks_compress(d_ins, [len(b) for b in blocks], outs, caps, 1)
    for k, b in enumerate(blocks):
        want = o.ans_encode(o.bwt_forward(b))
        got = outs[k][: n[k]].cpu().numpy()
        print(f"  2-block group, block {k}: {n[k]} vs {len(want)} bytes, equal {np.array_equal(got, want)}", flush=True)
