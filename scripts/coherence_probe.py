#!/usr/bin/env python3
"""What would ray ORDER buy the intersect kernel?  The same set of rays — per slot (four per pixel of a 1920x1080 image) the camera ray or a first / second / third bounce ray of a
random walk over the scene's surfaces, as a path pool holds them in steady state — goes through the production kernel (pt_debug_intersect, HIP-event time of the one
launch) in pixel order (what the pool's slot order resembles), shuffled, and sorted by direction octant and / or origin cell.  Hit records must not depend on the order.
usage: coherence_probe.py [config ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ptimport  # noqa: E402

pt = ptimport.load()
from pathtracer_0_amd import renderer, scenes  # noqa: E402

W, H, K = 1920, 1080, 4          # K slots per pixel: a launch of 8.3 M rays, what a steady-state iteration of the two-stream pool traces
N = W * H * K


def unit(v):
    return (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)


def morton3(c, bits):
    k = np.zeros(len(c), np.uint32)
    for b in range(bits):
        for a in range(3):
            k |= ((c[:, a] >> b) & 1).astype(np.uint32) << (3 * b + a)
    return k


for name in (sys.argv[1:] or ["C3", "C4", "C6"]):
    wl = scenes.build(name, W, H)
    r = renderer.Renderer(W, H)
    r.load_workload(wl)
    rs = np.random.RandomState(3)
    cam = np.asarray(wl.buffers[0], np.float32)
    # camera-like rays in pixel order (no lens model needed: a pinhole fan towards +z from the workload's origin)
    ys, xs = np.mgrid[0:H, 0:W]
    xs = np.repeat(xs.ravel(), K); ys = np.repeat(ys.ravel(), K)
    d0 = unit(np.stack([(xs / W - 0.5) * -1.5, (ys / H - 0.5) * 1.5 * H / W, np.ones(N)], axis=1))
    o = np.tile(cam, (N, 1)).astype(np.float32); d = d0
    bounce = rs.choice(4, size=N, p=[0.26, 0.26, 0.24, 0.24])          # ~3.9 segments per sample: every bounce depth about equally present
    O = o.copy(); D = d.copy()
    for b in range(1, 4):
        tuv, prim = r.debug_intersect(o, d)
        hit = prim >= 0
        t = np.where(hit, tuv[:, 0], 1.0).astype(np.float32)
        o = (o + (t[:, None] - 1e-3) * d).astype(np.float32)             # just in front of the surface (a miss: one unit along the ray)
        d = unit(rs.normal(size=(N, 3)))
        m = bounce >= b
        O[m] = o[m]; D[m] = d[m]
    lo, hi = O.min(0), O.max(0)
    cell = np.clip(((O - lo) / (hi - lo + 1e-6) * 16).astype(np.int64), 0, 15)
    octant = ((D[:, 0] < 0).astype(np.uint32) | ((D[:, 1] < 0).astype(np.uint32) << 1) | ((D[:, 2] < 0).astype(np.uint32) << 2))
    mort = morton3(cell, 4)
    orders = {"pixel order (the pool's slot order)": np.arange(N), "shuffled": rs.permutation(N), "by octant": np.argsort(octant, kind="stable"),
              "by origin cell (16^3, Morton)": np.argsort(mort, kind="stable"), "by octant, then cell": np.argsort(octant.astype(np.uint64) << 12 | mort, kind="stable"),
              "by cell, then octant": np.argsort(mort.astype(np.uint64) << 3 | octant, kind="stable")}
    # a sorted order hands whole regions of the scene to single blocks (the kernel deals each block one contiguous range): the same orders in chunks of 64 / 4096
    # consecutive rays, the chunks shuffled, keep the waves' coherence and balance the blocks
    for base in ("by cell, then octant", "by octant, then cell", "by origin cell (16^3, Morton)"):
        for chunk in (64, 4096):
            p = orders[base]; n = (N // chunk) * chunk
            rows = p[:n].reshape(-1, chunk)[rs.permutation(n // chunk)].reshape(-1)
            orders[f"{base}; chunks of {chunk} shuffled"] = np.concatenate([rows, p[n:]])
    ref = None
    print(f"{name}: {N} rays, bounce depths 0-3 mixed; one launch of the production intersect kernel each")
    for label, perm in orders.items():
        r.set_timing(True); r.reset_counters()
        best = None
        for rep in range(2):
            tuv, prim = r.debug_intersect(O[perm], D[perm])
            n, ms = r.kernel_time("extend")
            r.reset_counters()
            best = ms / max(n, 1) if best is None else min(best, ms / max(n, 1))
        inv = np.empty(N, np.int64); inv[perm] = np.arange(N)
        key = (tuv[inv].view(np.uint32), prim[inv])
        if ref is None:
            ref = key
        same = np.array_equal(ref[0], key[0]) and np.array_equal(ref[1], key[1])
        print(f"  {label:58s} {best:7.3f} ms   {N / best / 1e3:8.1f} Mrays/s   hit records equal to pixel order: {same}", flush=True)
    r.close()
