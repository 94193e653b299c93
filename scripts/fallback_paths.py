#!/usr/bin/env python3
"""Launches that USED to fall back to the compiled k_extend_persist (same results) and now run the hand-written kernel — with the rate on both kernels where both run:
  C4 on the hand-written kernel and forced onto the compiled one (the price of a fall-back on a scene both take),
  C3 and C5 with RAYTRACING = 0 (directDiffuse; C5 with the thickness probes of its subsurface materials: FL_PROBE rays set up at the hand-written kernel's refill since round 5),
  a one-object height field of ~1 M triangles (2 M nodes: 24-bit traversal-stack entries; until round 5 such trees ran the compiled kernel) on both kernels,
each 1920x1080, two streams on GPU 0, synchronous batches of 8 frames x SAMPLE_RES 8.  What still runs the compiled kernel: statistics runs, more than 1024 BVHs, empty leaves."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import ptimport  # noqa: E402

pt = ptimport.load()
from pathtracer_0_amd import hostlib, renderer, scenes  # noqa: E402
from bvh_bench import heightfield, obj_text  # noqa: E402

W, H, F = 1920, 1080, 8


def run(name, wl, **opts):
    r = renderer.Renderer(W, H, devices=[0, 0])
    for k, v in opts.items():
        r.set_option(k, v)
    r.load_workload(wl); r.reset_frame()
    seeds = [scenes.frame_seed(f) for f in range(1, F + 1)]
    r.render_batch(1, seeds); r.synchronize(); r.reset_frame()      # pool, rings
    t = time.perf_counter()
    for k in range(3):
        r.render_batch(1 + k * F, seeds)
    r.synchronize()
    dt = time.perf_counter() - t
    try:
        r.set_option("query_asm_launches_above", 0); kern = "pt_extend_asm"
    except renderer.PtError:
        kern = "k_extend_persist (compiled)"
    r.close()
    print(f"{name:64s} {W * H * wl.sample_res * F * 3 / dt / 1e6:8.1f} Msamples/s   intersect kernel: {kern}", flush=True)


c4 = scenes.build("C4", W, H)
run("C4, hand-written kernel", c4)
run("C4, compiled kernel (extend_mode 1)", c4, extend_mode=1)
run("C3, RAYTRACING = 0 (directDiffuse)", scenes.build("C3", W, H).with_params(RAYTRACING=0))
run("C5, RAYTRACING = 0 (directDiffuse + thickness probes)", scenes.build("C5", W, H).with_params(RAYTRACING=0))
v, f = heightfield(708, 708)
sc = hostlib.Scene()
sc.addMaterial("ground"); sc.setLastMtl("Kd", (0.7, 0.6, 0.5)); sc.setLastMtl("Pr", 1)
sc.use_gpu_bvh_builder(0)
sc.addObjectText(obj_text(v, f).encode(), 0)
big = scenes._finish("big", sc, W, H, (0.0, 0.8, -1.6), (0.35, 0.0, 0.0), (150, 180, 230), 8, 8)
run(f"height field, {len(f)} triangles / {big.info['nodes']} nodes, 8 bounces", big)
run("... forced onto the compiled kernel (extend_mode 1)", big, extend_mode=1)
