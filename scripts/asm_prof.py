#!/usr/bin/env python3
"""Where a wave of the hand-written intersect kernel spends its cycles (needs a -DPT_ASM_DEBUG -DPT_ASM_PROF -DPARK=0 build: PT_HIP_LIB, and
PT_ASM_DEBUG=1 in the environment): asm_prof.py [config] [frames] [option=value ...] -> per-wave accumulators of the LAST intersect launch."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ptimport  # noqa: E402

pt = ptimport.load()
from pathtracer_0_amd import renderer, scenes  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "C3"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 8
W, H = 1920, 1080
wl = scenes.build(name, W, H)
r = renderer.Renderer(W, H)
r.load_workload(wl); r.reset_frame(); r.set_option("path_slots", 1 << 23)
for kv in sys.argv[3:]:
    k, v = kv.split("="); r.set_option(k, int(v)); print("option", k, v)
# stop in the steady state: the last launch before the poll is a full one
r.render_batch_async(1, [scenes.frame_seed(f) for f in range(1, frames + 1)])
out = np.zeros(16 + 8192 * 8, np.uint64)
L = renderer.lib(); L.pt_debug_phase_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
L.pt_debug_phase_stats(r._h, out.ctypes.data, out.size)
w = out[16:].view(np.uint32).reshape(8192, 16)      # one record per BLOCK (the last of its waves to finish wrote it)
w = w[w[:, 0] != 0xffffffff].astype(np.float64)
fused = not any(a == "asm_loop=0" for a in sys.argv[3:]) and name != "C2"
names = (("loop overhead", "wait for the records", "node step", "triangle step", "wait for the pops", "requesting the next records", "next BVH / retire") if fused else
         ("votes+rest", "node fetch wait", "node step", "tri fetch wait", "tri step", "next BVH/retire", "-"))
life = w[:, 8].sum()
print(name, "fused trip" if fused else "phase-voting loop", "- waves sampled", len(w), " cycles per wave %.0f" % (life / max(len(w), 1)), " trips per wave %.1f" % w[:, 7].mean())
acc = 0.0
for k, nm in enumerate(names):
    t = w[:, k].sum(); acc += t
    print(f"  {nm:28s} {100 * t / life:5.1f} %   per trip: {t / max(w[:, 7].sum(), 1):7.1f} cycles")
print(f"  {'refills (the rest)':28s} {100 * (life - acc) / life:5.1f} %")
r.synchronize(); r.close()
