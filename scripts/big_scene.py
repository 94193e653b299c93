#!/usr/bin/env python3
"""End-to-end on a big single-object mesh (default 1 M triangles): OBJ parse, GPU BVH build, pack, upload + device re-layout, render,
oracle parity on a pixel lattice.  usage: big_scene.py [quads_x quads_y]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import ptimport  # noqa: E402

pt = ptimport.load()
from pathtracer_0_amd import hostlib, renderer, scenes  # noqa: E402
import oracle  # noqa: E402
from bvh_bench import heightfield, obj_text  # noqa: E402

qx, qy = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (708, 708)
W, H = 640, 360
t0 = time.perf_counter()
v, f = heightfield(qx, qy)
text = obj_text(v, f).encode()
t1 = time.perf_counter()
sc = hostlib.Scene()
sc.addMaterial("ground"); sc.setLastMtl("Kd", (0.7, 0.6, 0.5)); sc.setLastMtl("Pr", 1)
sc.use_gpu_bvh_builder(0)
sc.addObjectText(text, 0)
t2 = time.perf_counter()
wl = scenes._finish("big", sc, W, H, (0.0, 0.8, -1.6), (0.35, 0.0, 0.0), (150, 180, 230), 4, 4)
t3 = time.perf_counter()
r = renderer.Renderer(W, H)
r.load_workload(wl); r.reset_frame()
seeds = [scenes.frame_seed(1), scenes.frame_seed(2)]
r.render_batch(1, seeds[:1])
r.synchronize()
t4 = time.perf_counter()
r.render_batch(2, seeds[1:])
a = r.read_frame().copy()
t5 = time.perf_counter()
r.close()
osc = oracle.Scene.from_workload(wl)
ref = np.zeros((H, W, 4), np.float32)
for i, sd in enumerate(seeds):
    oracle.render(osc, W, H, 1 + i, sd, ref, nthreads=16, xs=8, ys=9)
ok = np.array_equal(a[::9, ::8], ref[::9, ::8])
print(f"{len(f)} triangles, {wl.info['nodes']} nodes, depth {wl.info['max_depth']}: mesh+OBJ text {t1 - t0:.2f} s | parse + GPU BVH {t2 - t1:.2f} s | pack {t3 - t2:.2f} s | "
      f"upload + device re-layout + first frame {t4 - t3:.2f} s | next frame {t5 - t4:.3f} s | oracle lattice parity: {ok}; mean {a[..., :3].mean() / 2:.4f}")
