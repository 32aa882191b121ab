"""The hand-scheduled GEMM issues LDS transpose reads as inline asm and waits for them a whole segment later: the compiler must not
touch a register such a read is still filling (tools/check_isa_async.py; the hazard class of the scheduler-fetch bug of round 3,
DESIGN_HISTORY.md s4).  Checked on the gfx950 code object of the in-tree build; skipped where the object or llvm-objdump is missing."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import check_isa_async as C  # noqa: E402


def test_checker_flags_a_register_touched_before_its_wait():
    asm = """
0000000000001000 <_ZN2v414gemm_v4_kernelILb0ELb1ELi0EEEvNS_5KArgsE>:
	ds_read_b64_tr_b16 v[10:11], v4 offset:1024          // 000000001000: AAAA
	v_mfma_f32_32x32x16_bf16 v[100:115], v[20:23], v[24:27], v[100:115]   // ok: other registers
	v_mov_b32_e32 v30, v11                               // 000000001010: copies a register that is still being filled
	s_waitcnt lgkmcnt(0)                                 // 000000001014
	v_mov_b32_e32 v31, v10                               // fine: after the wait
	ds_read_b64_tr_b16 v[12:13], v4 offset:2048
	s_waitcnt vmcnt(3)                                   // does not cover LDS reads
	v_add_u32_e32 v12, v12, v5                           // overwrites / reads a pending register
	s_endpgm
"""
    counts, violations = C.check(asm)
    assert list(counts.values()) == [2]
    assert [v[2] for v in violations] == [[11], [12]]


def test_no_asynchronous_read_is_touched_before_its_wait_in_the_built_gemm():
    obj = os.path.join(ROOT, "declip_amd", "csrc", "gemm_v4.o")
    if not (os.path.exists(obj) and os.path.exists(C.OBJDUMP)):
        pytest.skip("needs the in-tree build (__graft_entry__.build()) and llvm-objdump")
    counts, violations = C.check(C.disassemble(obj))
    assert sum(counts.values()) >= 5 * 64, counts            # the dX and dW flavours read their contraction-major operands this way
    assert not violations, violations[:5]
