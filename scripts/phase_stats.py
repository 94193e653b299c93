#!/usr/bin/env python3
"""Lane utilisation of k_extend_persist by phase (needs a library built with -DPT_PHASE_STATS, see PT_HIP_LIB)."""
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
cfg = scenes.CONFIGS[name]
W, H = (1920, 1080)
wl = scenes.build(name, W, H)
r = renderer.Renderer(W, H)
slots = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 23
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 16
r.load_workload(wl); r.reset_frame(); r.set_option("count_stats", 1); r.set_option("path_slots", slots)
for kv in sys.argv[4:]:
    k, v = kv.split("="); r.set_option(k, int(v)); print("option", k, v)
r.reset_counters()
r.render_batch(1, [scenes.frame_seed(f) for f in range(1, frames + 1)])
cnt = r.counters()            # (completes the batch)
out = np.zeros(16, np.uint64)
L = renderer.lib(); L.pt_debug_phase_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
L.pt_debug_phase_stats(r._h, out.ctypes.data, 16)
seg = cnt["segments"]
print(name, "pool", slots, "frames", frames, "segments", seg, "nodes/seg %.2f tritests/seg %.2f" % (cnt["nodes"] / seg, cnt["tritests"] / seg))
for k, nm in enumerate(("refill", "next-object/retire", "inner step", "leaf step", "outer loop")):
    trips, lanes = int(out[2 * k]), int(out[2 * k + 1])
    if trips:
        print(f"  {nm:20s} trips/seg x64 = {trips * 64 / seg:8.1f} lane-slots   active lanes/trip = {lanes / trips:5.1f} ({100 * lanes / trips / 64:4.1f} %)   active lane-trips/seg = {lanes / seg:6.2f}")
tt = [int(x) for x in out[10:15]]
if sum(tt):
    print("  wave time by phase (shader clock, lane 0 of every wave): " + "  ".join(f"{nm} {100.0 * t / sum(tt):.1f} %" for nm, t in zip(("refill", "next-object/retire", "inner", "leaf", "votes+rest"), tt))
          + f"   [{sum(tt) / seg:.0f} wave-cycles per segment]")
r.close()
