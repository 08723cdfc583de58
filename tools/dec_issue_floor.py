#!/usr/bin/env python3
"""are the two serial decode kernels at the issue floor of a lone wave?  Combines
   * the SQ counters of ONE decode of a 64 MiB text block (tools/pmc_sq.sh <dir> dec -> sq_summary.txt),
   * its wall time and symbol count (tools/dec_once.py -> decode_once.txt),
   * the per-instruction costs of one wave measured by tools/issuetest.hip (issuetest.txt)
   into profiles/<tag>_decode_issue_floor.txt:   python tools/dec_issue_floor.py gpurun_out/r05 > profiles/r05_decode_issue_floor.txt"""
import re
import sys

d = sys.argv[1]
rows = {}
hdr = None
for line in open(f"{d}/sq_dec/sq_summary.txt"):
    p = line.split()
    if hdr is None:
        hdr = p[1:p.index("(sums")] if "(sums" in p else p[1:]
        continue
    if len(p) >= len(hdr) + 1:
        name = " ".join(p[: len(p) - len(hdr)])
        rows[name] = dict(zip(hdr, map(float, p[len(p) - len(hdr):])))
once = open(f"{d}/decode_once.txt").read()
syms = int(re.search(r"(\d+) RLE0 symbols", once).group(1))
ms = float(re.search(r"([\d.]+) ms per decode", once).group(1))
it = {}
for line in open(f"{d}/issuetest.txt"):
    m = re.match(r"\s*(.+?)\s+([\d.]+) ns/instr\s+\(memtime ticks/instr ([\d.]+)\)", line)
    if m:
        it[m.group(1).strip()] = float(m.group(3))
print("# Decode chains against the issue rate of one wave (VERDICT r4 #6b).  One rANS decode of a 64 MiB enwik8-like block: 65 chunks = 65 waves,")
print(f"# {syms} RLE0 symbols ({syms / (64 << 20):.3f} per byte), {ms:.1f} ms wall (tools/dec_once.py); SQ counters of that one decode (tools/pmc_sq.sh dec;")
print("# SQ_WAVE_CYCLES / SQ_ACTIVE_* / SQ_WAIT_* count quad-cycles: x 4 = shader cycles); issue costs of one wave from tools/issuetest.hip on the same box.")
print()
print("issue costs of ONE wave (cycles per instruction, tools/issuetest.hip):")
for k in ("v_add_u32 x4 independent (per instr)", "v_add_u32 dependent", "add,xor dependent (per pair)", "add,xor,lshr,or dependent (per quad)",
          "v_cmp, s_ff1(vcc), v_readlane(sel), v_add (quad)", "readfirstlane + 3 salu + v_mov (5)", "s_cmp + NOT taken s_cbranch + 2 v_add (4 issued)",
          "s_cmp + TAKEN s_cbranch skip 1 + v_add (3 issued)", "ds_read_b32 dependent + wait", "global_load_dword dependent (L2/L1 hit)"):
    if k in it:
        n = {"(per pair)": 2, "(per quad)": 4, "(quad)": 4, "(5)": 5, "(4 issued)": 4, "(3 issued)": 3}
        div = next((v for s_, v in n.items() if k.endswith(s_)), 1)
        print(f"  {k:58s} {it[k]:7.2f} per group = {it[k] / div:5.2f} per instruction")
print()
for kern, unit in (("k_dec_rans", "RLE0 symbol (exponent + mantissa pair)"), ("k_dec_rank", "RLE0 symbol (= one run of the rank walk)")):
    r = rows.get(kern)
    if not r:
        continue
    valu, salu, lds, vm = r["SQ_INSTS_VALU"], r["SQ_INSTS_SALU"], r["SQ_INSTS_LDS"], r["SQ_INSTS_VMEM_RD"] + r["SQ_INSTS_VMEM_WR"]
    ins = valu + salu + lds + vm
    cyc = 4.0 * r["SQ_WAVE_CYCLES"]
    act = 4.0 * r["SQ_ACTIVE_INST_ANY"]
    wait = 4.0 * r["SQ_WAIT_ANY"]
    print(f"{kern}: per {unit}")
    print(f"  instructions  {ins / syms:6.1f}  (VALU {valu / syms:.1f}, SALU incl. branches {salu / syms:.1f}, LDS {lds / syms:.2f}, vector memory {vm / syms:.3f})")
    print(f"  wave cycles   {cyc / syms:6.1f}  (issuing {act / syms:.1f}, waiting {wait / syms:.1f}; the densest chunk sets the kernel's duration)")
    print(f"  => {cyc / ins:.2f} cycles per instruction;  at the best case of a lone wave (independent instructions, {it.get('v_add_u32 x4 independent (per instr)', 5.06):.2f}) the same instructions would take "
          f"{ins / syms * it.get('v_add_u32 x4 independent (per instr)', 5.06):.0f} cycles, as one dependent chain ({it.get('v_add_u32 dependent', 8.25):.2f}) {ins / syms * it.get('v_add_u32 dependent', 8.25):.0f}")
    print()
print("""Reading.  A lone wave issues one instruction every 5.1 cycles at best and one every 8.3 when each depends on the one before; a compare
-> scalar find-first -> readlane -> use sequence (the decoder's symbol search) costs 10 per instruction, a scalar instruction behind a vector
result 8, a not-taken branch 7, a taken one 13.  k_dec_rans runs its ~47 instructions per symbol at 6.5-6.8 cycles each and k_dec_rank its ~37 per
run at ~8.4 (three LDS round trips of 52-68 cycles per run are in it): both sit between the two lone-wave limits, i.e. the chains are
bound by the NUMBER of instructions on the dependent path, not by memory (vector memory: a store per 64 symbols; no LDS on k_dec_rans' common
path: its ds_bpermute live in the QuasiModel rebuild, 5.7e4 executions per block) and not by a stall that scheduling could remove -- the step
is state -> search -> next state, and the parked cycles (SQ_WAIT_ANY) are the interlocks behind its vector -> scalar hand-offs (the search quad
alone costs 40.6 cycles in isolation, 20 more than its four issue slots).  What is left inside one wave is the distance to the 5.1-cycle
limit: at most 22-26 % for k_dec_rans if every instruction issued back to back, which a dependent recurrence cannot do.
Two chunks interleaved in ONE wave (VERDICT r3 #6a) would fill those issue gaps -- exactly what a second resident wave on the same SIMD
already does: a SIMD issues one wave64 VALU instruction every 2 cycles, a chain needs ~29 of them per 300 cycles, so five chains saturate it,
and the batch decoder already places two to four chains per SIMD (4160 chains: k_dec_rans 142 ms against 137 for 65, k_dec_rank 149 against
131, profiles/r03_kernel_stats_decode_batch64.txt).  A batch is bound by its LONGEST chain (1.03 M symbols x ~305 cycles = 131 ms for each of the
two kernels), a single block by the same chain at one wave per SIMD; neither is helped by interleaving.  The lever that remains is fewer
instructions per symbol -- the round-2 work that took the symbol from 70 to 43 -- or a format with shorter chunks.""")
