"""Code-generation guard for the rANS encoder chain (k_rans_lanes, ans_enc.hip).

Round 4 made the chain's record prefetch correct by construction: the sixteen batches of records in flight land in an LDS ring
(global_load_lds_dwordx4 issued from asm volatile) and become register values only through an ordinary ds_read behind an asm
s_waitcnt -- no register ever holds data that is still in flight, so nothing the register allocator does (copies at the loop
back-edge, spills) can read stale data, which is what produced wrong bytes in round 3.  What is still hand-counted is
`s_waitcnt vmcnt(42)`: it is exact as long as the steady-state loop issues exactly ONE LDS-DMA load and TWO stores per batch (the
low halves of the kept states and, since round 6, the emit masks -- 14 batches x 3 between a load and the wait for it)
and all three come from the asm statements (anything the compiler adds is younger than the awaited load and only makes the wait
stricter; something it REMOVED or moved out of the loop would make it too weak).  This test compiles the translation unit to
gfx950 assembly and checks that census, and that no record is ever loaded into registers from global memory."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"
VMEM = re.compile(r"^\s*(global_|buffer_|scratch_|flat_)\w+")


def kernel_body():
    src = os.path.join(ROOT, "jampack_amd", "csrc", "ans_enc.hip")
    out = "/tmp/jpk_ans_enc_codegen.s"
    subprocess.check_call([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-DJPK_BUILD", "--cuda-device-only", "-S", src, "-o", out],
                          stderr=subprocess.DEVNULL)
    text = open(out).read()
    m = re.search(r"^(_ZN\S*k_rans_lanes\S*):", text, re.M)
    assert m, "k_rans_lanes not found in the device assembly"
    return text[m.start(): text.index("s_endpgm", m.start())].splitlines()


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not present")
def test_the_encoder_chain_loop_issues_exactly_two_stores_and_one_load_per_batch():
    body = kernel_body()
    waits = [i for i, ln in enumerate(body) if re.search(r"s_waitcnt\s+vmcnt\(42\)", ln)]
    assert len(waits) == 16, f"expected the sixteen-fold unrolled steady-state loop, found {len(waits)} waits"
    # the loop: a label above the first wait that a branch below the last wait jumps back to
    label = None
    for i in range(waits[0], -1, -1):
        m = re.match(r"^(\.LBB\d+_\d+):", body[i])
        if m and any(re.search(r"s_cbranch_\w+\s+" + re.escape(m.group(1)) + r"\b", ln) for ln in body[waits[-1]:]):
            label = i
            break
    assert label is not None, "the loop around the steady-state waits was not recognised"
    # census of every batch: from its wait to the next wait (the last one: to the back edge) exactly the two stores followed by one
    # LDS-DMA record load, nothing else that counts on vmcnt; the record is read from LDS behind the wait; sixteen steps
    bounds = waits + [max(j for j, ln in enumerate(body) if re.search(r"s_cbranch_\w+\s+" + re.escape(body[label].split(":")[0]) + r"\b", ln))]
    for a, b in zip(bounds[:-1], bounds[1:]):
        seg = body[a:b]
        vmem = [ln.strip().split()[0] for ln in seg if VMEM.match(ln)]
        assert vmem == ["global_store_short", "global_store_dword", "global_load_lds_dwordx4"], (a, vmem)
        assert sum(1 for ln in seg if "ds_read_b128" in ln) == 1, "one LDS read of the next record per batch"
    # sixteen steps per batch (the compiler may rotate a batch's first steps in front of its wait: they run on `cur`, landed data)
    assert sum(1 for ln in body[label: bounds[-1]] if "row_ror:1" in ln) == 16 * 16
    # nothing that counts on vmcnt between the head of the loop and the first wait either
    assert not [ln for ln in body[label: waits[0]] if VMEM.match(ln)]


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not present")
def test_no_record_is_loaded_into_registers_and_m0_is_ours():
    body = kernel_body()
    assert not [ln for ln in body if re.search(r"global_load_dwordx4\s+v", ln)], "a record load with a register destination is back"
    assert not [ln for ln in body if re.match(r"^\s*scratch_", ln)], "the chain kernel spills"
    # sixteen prologue loads + sixteen in the loop, all LDS-DMA; drained before the loop and before the wave releases its LDS
    assert sum(1 for ln in body if "global_load_lds_dwordx4" in ln) == 32
    assert sum(1 for ln in body if re.search(r"s_waitcnt\s+vmcnt\(0\)", ln)) >= 3
    # M0 (the LDS address of an LDS-DMA load) is compiler-reserved and NOT saved around our statements: nothing else in the kernel
    # may touch it -- every mention is our `s_mov_b32 m0, <slot>` and each is followed by its load within three instructions
    mentions = [i for i, ln in enumerate(body) if re.search(r"\bm0\b", ln) and not ln.strip().startswith(";")]
    assert len(mentions) == 32
    for i in mentions:
        assert re.match(r"^\s*s_mov_b32 m0, s\d+", body[i]), body[i]
        nxt = [ln.strip().split()[0] for ln in body[i + 1: i + 5] if ln.strip() and not ln.strip().startswith(";")]
        assert "global_load_lds_dwordx4" in nxt[:3], (body[i], nxt)
