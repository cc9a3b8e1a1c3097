"""CPU: the hand-counted operand rings of csrc/lstm_coop.hip (inline-asm requests + `s_waitcnt vmcnt(N)` written by hand) rest
on the register allocator never touching a request's destination registers while the load is in flight -- which the compiler
does not know about.  tools/check/asm_rings.py replays the vector-memory queue over the generated ISA and must find nothing
(round 4: a conv operand ring built the same way was copied out of its registers in front of the wait and was dropped)."""
import importlib.util
import os
import shutil

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool():
    spec = importlib.util.spec_from_file_location("asm_rings", os.path.join(ROOT, "tools", "check", "asm_rings.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_no_instruction_touches_a_ring_register_in_flight():
    t = _tool()
    findings, kernels = t.check(t.isa_of(os.path.join(t.CSRC, "lstm_coop.hip")))
    assert kernels >= 2, kernels                       # the K-split backward kernels hold the rings
    assert not findings, findings[:10]


def test_the_checker_sees_a_copy_in_front_of_the_wait(tmp_path):
    t = _tool()
    isa = tmp_path / "k.s"
    isa.write_text("""_Zk:
	;;#ASMSTART
	global_load_dwordx4 v[4:7], v[2:3], off
	;;#ASMEND
	;;#ASMSTART
	global_load_dwordx4 v[8:11], v[2:3], off
	;;#ASMEND
	v_mov_b64_e32 v[20:21], v[8:9]
	;;#ASMSTART
	s_waitcnt vmcnt(1)
	;;#ASMEND
	ds_write_b128 v30, v[4:7]
	ds_write_b128 v30, v[8:11]
	;;#ASMSTART
	s_waitcnt vmcnt(0)
	;;#ASMEND
	ds_write_b128 v30, v[8:11]
.Lfunc_end0:
""")
    findings, kernels = t.check(str(isa))
    assert kernels == 1
    assert [f[2].split()[0] for f in findings] == ["v_mov_b64_e32", "ds_write_b128"] and "v[8:11]" in findings[1][2]
