#!/usr/bin/env python3
"""The hand-written intersect kernel under contention: while two more contexts render C3 on the same GPU from threads of their own, this process traces the
same random rays again and again on the production kernel (block shape of the shared-GPU mode) and compares every hit record with the compiled
kernel's.  pt_debug_intersect zeroes the hit records first, so a record a launch fails to write shows up as a difference, not as a fault.
    contention_check.py [config] [rays] [repeats] [option=value ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ptimport  # noqa: E402

pt = ptimport.load()
from pathtracer_0_amd import renderer  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "C3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 18
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
opts = dict(a.split("=") for a in sys.argv[4:])
W, H = 96, 54
wl = pt.scenes.build(name, W, H)
rs = np.random.RandomState(5)
o = (np.array(wl.buffers[0]) + rs.normal(scale=0.3, size=(n, 3))).astype(np.float32)
d = rs.normal(size=(n, 3)).astype(np.float32)
d /= np.linalg.norm(d, axis=1, keepdims=True)
r = renderer.Renderer(W, H)
r.load_workload(wl)
for k, v in {"asm_tpb": 256, "extend_blocks_per_cu": 6, "extend_cache_bytes": 16384, **opts}.items():
    r.set_option(k, int(v))
r.set_option("extend_mode", 1)
tuv, prim = r.debug_intersect(o, d)
ref = (tuv.view(np.uint32).copy(), prim.copy())
r.set_option("extend_mode", 2)
import threading  # noqa: E402
stop = False


def load():
    """two more streams rendering C3 on the same GPU, each from a thread of its own, until told to stop"""
    rr = renderer.Renderer(960, 540)
    rr.load_workload(pt.scenes.build("C3", 960, 540)); rr.reset_frame()
    f = 1
    while not stop:
        rr.render_batch(f, [pt.scenes.frame_seed(f + k) for k in range(4)]); f += 4
    rr.close()


threads = [threading.Thread(target=load) for _ in range(2)]
for t in threads:
    t.start()
import time  # noqa: E402
time.sleep(1.0)
bad = 0
for k in range(reps):
    tuv, prim = r.debug_intersect(o, d)
    same = (prim == ref[1]) & (tuv.view(np.uint32) == ref[0]).all(axis=1)
    if not same.all():
        idx = np.nonzero(~same)[0]
        bad += 1
        print(f"repeat {k}: {idx.size} of {n} records differ; first {idx[:8]}; groups of 64: {np.unique(idx // 64)[:12]}; got prim {prim[idx[:6]]} t {tuv[idx[:6], 0]}  expected prim {ref[1][idx[:6]]} t {ref[0].view(np.float32)[idx[:6], 0]}", flush=True)
    if k % 10 == 9:
        print(f"{k + 1} repeats, {bad} bad", flush=True)
stop = True
for t in threads:
    t.join()
r.set_option("query_asm_launches_above", reps - 1)
r.close()
print(f"{name} rays {n} repeats {reps}: {bad} launches with differences")
sys.exit(1 if bad else 0)
