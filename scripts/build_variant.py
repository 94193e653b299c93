#!/usr/bin/env python3
"""A/B builds of libpt_hip.so into build/ab/<name>.so (run them with PT_HIP_LIB or scripts/ab.sh):
    build_variant.py name [-DASM_SWITCH=0 ...] [--hip -DFLAG ...]
-D flags before --hip go to the assembler (pt_extend_gfx950.s), those after it to hipcc."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ptimport  # noqa: E402

ptimport.load()
from pathtracer_0_amd import build  # noqa: E402

name, rest = sys.argv[1], sys.argv[2:]
asm = rest[:rest.index("--hip")] if "--hip" in rest else rest
hip = rest[rest.index("--hip") + 1:] if "--hip" in rest else []
out_dir = os.path.join(ROOT, "build", "ab")
os.makedirs(out_dir, exist_ok=True)
inc = os.path.join(out_dir, f"{name}_hsaco.inc")
build.assemble_extend(out_inc=inc, defines=asm)
d = os.path.join(build.HERE, "csrc", "hip")
out = os.path.join(out_dir, f"{name}.so")
subprocess.check_call(["/opt/rocm/bin/hipcc"] + build.HIP_FLAGS + hip + [f'-DPT_EXTEND_INC="{inc}"', "-o", out, os.path.join(d, "pt_hip.hip"), os.path.join(d, "pt_bvh.hip")])
print(out)
