"""Code-generation guard for the rANS encoder chain (k_rans_lanes, ans_enc.hip).

The chain keeps sixteen record loads in flight across the back-edge of its loop: the loads are issued from inline asm, the compiler
believes their destination registers hold the records from that statement on, and a register-to-register copy of such a tuple that
it inserts before the covering s_waitcnt (for instance at the loop back-edge, when the allocation of the loop-carried tuples
differs between the end and the head of the loop) copies whatever the registers held BEFORE the data arrived.  Round 3 hit
exactly that after an edit of the step (rare, timing-dependent wrong bytes on long chunks).  This test compiles the translation
unit to gfx950 assembly and refuses any vector move whose source is a destination of the kernel's in-flight record loads."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not present")
def test_record_tuples_of_the_encoder_chain_are_never_copied(tmp_path):
    src = os.path.join(ROOT, "jampack_amd", "csrc", "ans_enc.hip")
    out = tmp_path / "ans_enc.s"
    subprocess.check_call([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-DJPK_BUILD", "--cuda-device-only", "-S", src, "-o", str(out)],
                          stderr=subprocess.DEVNULL)
    text = open(out).read()
    m = re.search(r"^(_ZN\S*k_rans_lanes\S*):.*?^\s*\.end_amdhsa_kernel|^(_ZN\S*k_rans_lanes\S*):.*?s_endpgm", text, re.S | re.M)
    assert m, "k_rans_lanes not found in the device assembly"
    start = text.index(m.group(1) or m.group(2))
    body = text[start: text.index("s_endpgm", start)]
    loaded = set()
    for a, b in re.findall(r"global_load_dwordx4 v\[(\d+):(\d+)\]", body):
        loaded.update(range(int(a), int(b) + 1))
    assert len(loaded) >= 64, "the sixteen record tuples were not recognised"

    def regs(tok):
        t = re.match(r"v\[(\d+):(\d+)\]", tok)
        if t:
            return set(range(int(t.group(1)), int(t.group(2)) + 1))
        t = re.match(r"v(\d+)$", tok)
        return {int(t.group(1))} if t else set()

    bad = []
    for ln in body.splitlines():
        ln = ln.strip()
        mm = re.match(r"(v_mov_b32_e32|v_mov_b64_e32|v_pk_mov_b32|v_accvgpr_write_b32|v_mov_b32_e64)\s+(\S+),\s*(\S+)", ln)
        if mm and regs(mm.group(3).rstrip(",")) & loaded:
            bad.append(ln)
    assert not bad, "record tuples with loads in flight are copied between registers:\n" + "\n".join(bad[:10])
