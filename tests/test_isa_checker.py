"""tools/check_opaque_loads.py (the build-time guard of the hand-waited prefetch loads in csrc/det_mm.hip) on hand-written ISA snippets:
it must flag a compiler-generated read / copy / spill / overwrite of a destination register between an opaque load and its wait, follow
forward branches, resolve hipcc's flag-register lowering of a uniform if / else, and stay silent on the legal patterns."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("check_opaque_loads", os.path.join(ROOT, "tools", "check_opaque_loads.py"))
chk = importlib.util.module_from_spec(spec)
spec.loader.exec_module(chk)

HEAD = ["_Z6kernelv:", "\ts_mov_b32 s0, 0", ".LBB0_1:"]
LOAD = ["\t;;#ASMSTART", "\tglobal_load_dwordx4 v[4:7], v[2:3], off", "\t;;#ASMEND"]
WAIT = ["\t;;#ASMSTART", "\ts_waitcnt vmcnt(0)", "\t;;#ASMEND"]
TAIL = ["\ts_cbranch_scc1 .LBB0_1", "\ts_endpgm"]


def run(body):
    bad, nk = chk.check(HEAD + body + TAIL)
    return bad, nk


def test_clean_pattern_passes():
    bad, nk = run(LOAD + ["\tv_add_u32_e32 v8, v9, v10"] + WAIT + ["\tv_mov_b32_e32 v1, v4"])
    assert nk == 1 and not bad


def test_read_copy_spill_and_overwrite_are_flagged():
    for ins in ("\tv_mov_b32_e32 v20, v5", "\tscratch_store_dword off, v6, s0", "\tv_add_u32_e32 v4, v9, v10", "\tv_accvgpr_write_b32 a0, v7"):
        bad, _ = run(LOAD + [ins] + WAIT)
        assert len(bad) == 1 and bad[0][2] == ins.strip(), (ins, bad)


def test_forward_branch_carries_the_in_flight_set():
    body = LOAD + ["\ts_cbranch_vccz .LBB0_2", "\tv_add_u32_e32 v8, v9, v10", ".LBB0_2:", "\tv_mov_b32_e32 v20, v6"] + WAIT
    bad, _ = run(body)
    assert len(bad) == 1 and bad[0][3] == [6]


def test_compiler_vmcnt0_clears_the_set():
    bad, _ = run(LOAD + ["\ts_waitcnt vmcnt(0)", "\tv_mov_b32_e32 v20, v6"])
    assert not bad


def test_uniform_if_else_through_a_flag_register_is_resolved():
    # then side issues the load and clears the flag; the else side (which re-uses v4) is only reachable with the flag still set
    body = ["\ts_mov_b64 s[0:1], -1", "\ts_cbranch_vccnz .LBB0_3"] + LOAD + ["\ts_mov_b64 s[0:1], 0", ".LBB0_3:",
            "\ts_andn2_b64 vcc, exec, s[0:1]", "\ts_cbranch_vccnz .LBB0_4", "\tv_mov_b32_e32 v4, 0", ".LBB0_4:"] + WAIT
    bad, _ = run(body)
    assert not bad, bad
    # without the flag bookkeeping (flag overwritten by something else) the walk must stay conservative
    body2 = [x if "s_mov_b64 s[0:1], 0" not in x else "\ts_and_b64 s[0:1], s[2:3], s[4:5]" for x in body]
    bad2, _ = run(body2)
    assert len(bad2) == 1


def test_unreached_kernel_is_an_error():
    txt = ["_Z6kernelv:", "\ts_branch .LBB0_9", "\ts_mov_b32 s0, 0"] + LOAD + WAIT + [".LBB0_9:", "\ts_endpgm"]
    bad, nk = chk.check(txt)
    assert nk == 0 and len(bad) == 1 and "never reached" in bad[0][2]
