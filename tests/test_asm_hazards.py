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
    assert len(objs) == 8
    for f in objs:
        n, found = _checker().scan(os.path.join(d, f))
        assert n > 1000 and not found, (f, found[:5])
