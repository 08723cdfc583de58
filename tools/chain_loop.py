import os, sys, time
sys.path.insert(0, '/root/repo')
import torch
import jampack_amd as jam
n = 64 << 20
t = jam.corpus.make("text_survey", n, 8)
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream()
ctx = jam.Context(0, st.cuda_stream)
d_in = torch.from_numpy(t).to(dev)
cap = jam.ans_capacity(n + 480)
d_enc = torch.empty(cap, dtype=torch.uint8, device=dev)
for mode in (0, 2):
    ctx.profile_enable(mode)
    for rep in range(6):
        t0 = time.perf_counter()
        ctx.block_compress(d_in, n, d_enc, cap)
        dt = time.perf_counter() - t0
        s = ctx.stats()
        print(f"prof={mode} rep {rep}: compress {dt*1e3:.1f} ms; slowest chain {s.enc_chain_cycles / max(s.enc_chain_steps, 1):.1f} cyc/step, {s.enc_chain_ns / 1e6:.2f} ms, clock {s.enc_chain_cycles / max(s.enc_chain_ns, 1):.3f} GHz")
    if mode:
        for r in ctx.profile_table():
            if 'rans_lanes' in r['name']: print(r)
