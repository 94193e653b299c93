#!/usr/bin/env python3
"""When do the waves of one intersect launch finish?  (developer build -DPT_PHASE_STATS; PT_HIP_LIB)"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ptimport  # noqa: E402

pt = ptimport.load()
from pathtracer_0_amd import renderer, scenes  # noqa: E402

W, H = 1920, 1080
slots = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 23
wl = scenes.build("C3", W, H)
r = renderer.Renderer(W, H)
r.load_workload(wl); r.reset_frame(); r.set_option("path_slots", slots)
for kv in sys.argv[2:]:
    k, v = kv.split("="); r.set_option(k, int(v)); print("option", k, v)
L = renderer.lib(); L.pt_debug_phase_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
out = np.zeros(16 + 2 * 8192, np.uint64)
r.render_batch_async(1, [scenes.frame_seed(f) for f in range(1, 33)])      # returns with the pool still full: the last launch was a steady-state one
L.pt_debug_phase_stats(r._h, out.ctypes.data, 16 + 2 * 8192)
te = out[16:16 + 8192].astype(np.int64)
ts = out[16 + 8192:].astype(np.int64)
ts = ts[te > 0]; te = te[te > 0]
t0 = ts.min()
print('wave START times after the first start (us), percentiles:', {p: round(float(np.percentile(ts, p) - t0) / 100, 1) for p in (0, 25, 50, 75, 76, 90, 99, 100)})
print('wave DURATION (us), percentiles:', {p: round(float(np.percentile(te - ts, p)) / 100, 1) for p in (0, 1, 10, 50, 90, 99, 100)})
print('launch span (us):', float(te.max() - t0) / 100)
pc = [0, 1, 10, 50, 90, 99, 99.9]
print("pool", slots, ":", len(te), "waves; how long before the last wave the p-th percentile wave finished (us):")
print("  ", {p: round(float(te.max() - np.percentile(te, p)) / 100.0, 1) for p in pc})
allw = out[16:16 + 8192].astype(np.int64)
blk = np.arange(8192) // 8
for x in range(8):                                                   # blocks go round-robin over the 8 XCDs; their 100 MHz counters need not agree
    t = allw[(blk % 8 == x) & (allw > 0)]
    print(f"   XCD {x}: {len(t)} waves, last-first {float(t.max() - t.min()) / 100:.1f} us;", {p: round(float(t.max() - np.percentile(t, p)) / 100.0, 1) for p in (10, 50, 90, 99)},
          "offset of its last wave vs global last", round(float(allw.max() - t.max()) / 100, 1))
r.close()

dur = (out[16:16 + 8192].astype(np.int64) - out[16 + 8192:].astype(np.int64)) / 100.0
w = np.arange(8192)
print("mean duration (us) by XCD (block % 8):", [round(float(dur[(w // 8) % 8 == x].mean()), 1) for x in range(8)])
print("mean duration by wave-in-block:", [round(float(dur[w % 8 == k].mean()), 1) for k in range(8)])
print("mean duration by block index quartile:", [round(float(dur[(w // 8) // 256 == q].mean()), 1) for q in range(4)])
print("mean duration by slot range decile (wave id):", [round(float(dur[w // 820 == q].mean()), 1) for q in range(10)])
