"""The hand-written intersect kernel carries its own s_nops: scripts/asm_hazards.py scans the assembled code objects for the gfx940-family
hazards LLVM's hazard recognizer would have padded in compiled code (VALU-written SGPR / VCC read by the next VALU, v_readfirstlane of a
just-written VGPR, v_div_fmas after a VCC write, ...).  CPU test: the assembler cross-assembles without a GPU."""
import importlib.util
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def _checker():
    spec = importlib.util.spec_from_file_location("asm_hazards", os.path.join(ROOT, "scripts", "asm_hazards.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    return m


def _assemble(tmp_path, body):
    src = tmp_path / "t.s"
    src.write_text('.amdgcn_target "amdgcn-amd-amdhsa--gfx950"\n.text\nt:\n' + body + "\n  s_endpgm\n")
    obj, co = tmp_path / "t.o", tmp_path / "t.hsaco"
    subprocess.check_call([os.path.join(LLVM, "clang"), "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", str(src), "-o", str(obj)])
    subprocess.check_call([os.path.join(LLVM, "ld.lld"), "-shared", str(obj), "-o", str(co)])
    return str(co)


@pytest.mark.skipif(not os.path.exists(os.path.join(LLVM, "clang")), reason="no ROCm LLVM")
@pytest.mark.parametrize("body,rules", [
    ("  v_cmp_lt_f32_e32 vcc, v1, v2\n  v_cndmask_b32_e32 v3, v4, v5, vcc", {"A"}),
    ("  v_cmp_lt_f32_e32 vcc, v1, v2\n  s_nop 1\n  v_cndmask_b32_e32 v3, v4, v5, vcc", set()),
    ("  v_cmp_lt_f32_e64 s[4:5], v1, v2\n  s_nop 0\n  v_cndmask_b32_e64 v3, v4, v5, s[4:5]", {"A"}),
    ("  v_mov_b32_e32 v1, v2\n  v_readfirstlane_b32 s4, v1", {"E"}),
    ("  v_mov_b32_e32 v1, v2\n  s_nop 0\n  v_readfirstlane_b32 s4, v1", set()),
    ("  v_cmp_lt_f32_e32 vcc, v1, v2\n  s_nop 1\n  v_div_fmas_f32 v3, v4, v5, v6", {"C"}),
    ("  v_readfirstlane_b32 s4, v1\n  s_nop 2\n  global_load_dword v2, v3, s[4:5]", {"D"}),
    ("  v_rcp_f32_e32 v1, v2\n  v_mul_f32_e32 v3, v1, v1", {"F"}),
    ("  global_store_dwordx4 v1, v[4:7], s[4:5]\n  v_mov_b32_e32 v5, 0", {"G"}),
    ("  v_readfirstlane_b32 s4, v1\n  s_nop 1\n  v_readlane_b32 s5, v2, s4", {"B"}),
    # a scalar load still in flight when a branch leaves for code that reuses its registers (the round-3 refill bug: root boxes into s[64:71], the trip's masks)
    ("  s_load_dwordx2 s[4:5], s[0:1], 0x0\n  s_cbranch_execz skip\n  s_waitcnt lgkmcnt(0)\n  s_mov_b32 s6, s4\nskip:\n  s_mov_b64 s[4:5], exec", {"S"}),
    ("  s_load_dwordx2 s[4:5], s[0:1], 0x0\n  s_waitcnt lgkmcnt(0)\n  s_cbranch_execz skip\n  s_mov_b32 s6, s4\nskip:\n  s_mov_b64 s[4:5], exec", set()),
    # the same hazards across a TAKEN branch and across a loop back-edge (look-back runs over every predecessor of a label)
    ("  v_cmp_lt_f32_e32 vcc, v1, v2\n  s_cbranch_scc1 L\n  s_nop 3\nL:\n  v_cndmask_b32_e32 v3, v4, v5, vcc", {"A"}),
    ("  v_cmp_lt_f32_e32 vcc, v1, v2\n  s_nop 0\n  s_cbranch_scc1 L\n  s_nop 3\nL:\n  v_cndmask_b32_e32 v3, v4, v5, vcc", set()),
    ("L:\n  v_cndmask_b32_e32 v3, v4, v5, vcc\n  s_nop 3\n  v_cmp_lt_f32_e32 vcc, v1, v2\n  s_cbranch_scc1 L", {"A"}),
    ("  v_mov_b32_e32 v1, v2\n  s_branch L\n  s_nop 0\nL:\n  v_readfirstlane_b32 s4, v1", set()),                 # the branch itself is a wait state
    ("  v_readfirstlane_b32 s4, v1\n  s_cbranch_scc0 L\n  s_nop 4\nL:\n  global_load_dword v2, v3, s[4:5]\n  s_waitcnt vmcnt(0)", {"D"}),
    # a scalar load whose registers are reused 450 instructions later without a wait (no depth limit on the walk)
    ("  s_load_dwordx2 s[4:5], s[0:1], 0x0\n" + "  s_nop 0\n" * 450 + "  s_mov_b32 s6, s4", {"S"}),
    # V: vector-memory loads / LDS reads whose destination is touched before a wait that covers it
    ("  global_load_dword v1, v2, s[4:5]\n  v_mov_b32_e32 v3, v1\n  s_waitcnt vmcnt(0)", {"V"}),
    ("  global_load_dword v1, v2, s[4:5]\n  v_mov_b32_e32 v1, 0\n  s_waitcnt vmcnt(0)", {"V"}),                         # written, not read: the data lands on top
    ("  global_load_dword v1, v2, s[4:5]\n  global_load_dword v5, v2, s[4:5] offset:4\n  s_waitcnt vmcnt(1)\n  v_mov_b32_e32 v3, v1\n  s_waitcnt vmcnt(0)", set()),
    ("  global_load_dword v1, v2, s[4:5]\n  global_load_dword v5, v2, s[4:5] offset:4\n  s_waitcnt vmcnt(1)\n  v_mov_b32_e32 v3, v5\n  s_waitcnt vmcnt(0)", {"V"}),
    ("  ds_read_b32 v1, v2\n  s_waitcnt vmcnt(0)\n  v_mov_b32_e32 v3, v1\n  s_waitcnt lgkmcnt(0)", {"V"}),               # the wrong counter
    ("  ds_read_b32 v1, v2\n  ds_write_b32 v2, v6\n  s_waitcnt lgkmcnt(1)\n  v_mov_b32_e32 v3, v1\n  s_waitcnt lgkmcnt(0)", set()),   # LDS operations return in order
    # a load left in flight across the back-edge and used at the top of the loop before the wait (the fetch-at-decision pattern gone wrong)
    ("L:\n  v_add_u32_e32 v3, v1, v1\n  s_waitcnt vmcnt(0)\n  global_load_dword v1, v2, s[4:5]\n  s_cbranch_scc1 L\n  s_waitcnt vmcnt(0)", {"V"}),
    ("L:\n  s_waitcnt vmcnt(0)\n  v_add_u32_e32 v3, v1, v1\n  global_load_dword v1, v2, s[4:5]\n  s_cbranch_scc1 L\n  s_waitcnt vmcnt(0)", set()),
    # one path of two waits, the other does not
    ("  global_load_dword v1, v2, s[4:5]\n  s_cbranch_scc1 L\n  s_waitcnt vmcnt(0)\nL:\n  v_mov_b32_e32 v3, v1\n  s_waitcnt vmcnt(0)", {"V"}),
    # the two arms of a fetch: disjoint lanes of the same registers from two counters, one wait for both (the kernel's NODE_LDS / NODE_GLB)
    ("  ds_read_b32 v1, v2\n  global_load_dword v1, v3, s[4:5]\n  s_waitcnt vmcnt(0) lgkmcnt(0)\n  v_mov_b32_e32 v4, v1", set()),
    # a path the scan cannot follow is a finding, not a silent pass
    ("  s_getpc_b64 s[4:5]\n  s_setpc_b64 s[4:5]", {"X"}),
])
def test_checker_finds_what_it_is_meant_to(tmp_path, body, rules):
    _, found = _checker().scan(_assemble(tmp_path, body))
    assert {f[0] for f in found} == rules, found


@pytest.mark.skipif(not os.path.exists(os.path.join(LLVM, "clang")), reason="no ROCm LLVM")
def test_product_code_objects_are_hazard_free(pt):
    from pathtracer_0_amd import build
    build.assemble_extend()
    d = os.path.join(ROOT, "build", "asm", "pt_extend_hsaco")
    objs = sorted(f for f in os.listdir(d) if f.endswith(".hsaco"))
    assert len(objs) == 18
    for f in objs:
        n, found, waived = _checker().scan(os.path.join(d, f), waive=True)
        assert n > 1000 and not found, (f, found[:5])
        # nothing rides on a waiver (round 4 waived the node step's pop into vCur against the triangle step's advance of vCur as lane-disjoint; the pop now
        # lands in a register of its own and moves to vCur behind the wait)
        assert _checker().WAIVERS == [] and waived == [], (f, waived)
