#!/usr/bin/env python3
"""Developer check of the hand-written intersect kernel against the compiled ones on random rays: asm_check.py [config] [n]
prints where (which 64-ray groups) and how the hit records differ."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ptimport  # noqa: E402

pt = ptimport.load()
from pathtracer_0_amd import renderer  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "C2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
W, H = 96, 54
wl = pt.scenes.build(name, W, H)
rs = np.random.RandomState(3)
o = (np.array(wl.buffers[0]) + rs.normal(scale=0.3, size=(n, 3))).astype(np.float32)
d = rs.normal(size=(n, 3)).astype(np.float32)
d /= np.linalg.norm(d, axis=1, keepdims=True)
r = renderer.Renderer(W, H)
r.load_workload(wl)
for k, v in [a.split("=") for a in sys.argv[3:]]:
    r.set_option(k, int(v))
out = {}
for mode in (1, 2):
    r.set_option("extend_mode", mode)
    tuv, prim = r.debug_intersect(o, d)
    out[mode] = (tuv.copy(), prim.copy())
try:
    r.set_option("query_asm_eligible", 0); print("hand-written kernel takes this scene")
except Exception as e:
    print("NOT taken:", e)
bad = ~((out[1][1] == out[2][1]) & (out[1][0].view(np.uint32) == out[2][0].view(np.uint32)).all(axis=1))
print(name, "rays", n, "differing", int(bad.sum()), "hits (compiled)", int((out[1][1] >= 0).sum()))
if bad.any():
    idx = np.nonzero(bad)[0]
    groups = np.unique(idx // 64)
    print("64-ray groups with differences:", groups[:40], "of", n // 64)
    for i in idx[:12]:
        print(i, "compiled", out[1][0][i], out[1][1][i], " asm", out[2][0][i], out[2][1][i])
r.close()
