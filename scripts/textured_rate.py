#!/usr/bin/env python3
"""What texture-mapped materials cost: C3 (1920x1080, two streams, synchronous batches of 8 frames x 8 spp) as it is, and with a 1024x1024 map_Kd on every material
(k_shade's TEX variant: mapMtl + the software bilinear sampler; the map is white, so the paths and the image are C3's, and the triangles of C3 carry no vt, so every
lookup lands on one texel — the kernel variant and its registers are what is measured, not texture bandwidth), and T1 (the parity workload with uv-mapped quads and six kinds of maps) at the same size."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import ptimport  # noqa: E402

pt = ptimport.load()
from pathtracer_0_amd import renderer, scenes  # noqa: E402

W, H, F = 1920, 1080, 8


def run(name, wl):
    r = renderer.Renderer(W, H, devices=[0, 0])
    r.load_workload(wl); r.reset_frame()
    seeds = [scenes.frame_seed(f) for f in range(1, F + 1)]
    r.render_batch(1, seeds); r.synchronize(); r.reset_frame()
    t = time.perf_counter()
    for k in range(3):
        r.render_batch(1 + k * F, seeds)
    r.synchronize()
    dt = time.perf_counter() - t
    r.close()
    print(f"{name:60s} {W * H * wl.sample_res * F * 3 / dt / 1e6:8.1f} Msamples/s", flush=True)


c3 = scenes.build("C3", W, H)
run("C3", c3)
b = dict(c3.buffers); m = np.asarray(b[14], np.float32).copy(); me = int(m[0])
for k in range((len(m) - 1) // me):
    m[me * k + 23] = 1.0                               # map_Kd = texture 1 (slot 23 of the 48-float record, dispatch.java:295-315)
b[14] = m
tex = {1: np.full((1024, 1024, 4), 255, np.uint8)}      # white: Kd * 1.0 = Kd, the same paths and the same image — what differs is the kernel variant and the lookups
run("C3 + a white 1024x1024 map_Kd on every material", scenes.Workload("C3tex", W, H, b, c3.sky, c3.sample_res, c3.max_bounces, dict(c3.info), tex))
run("T1 (uv-mapped quads, map_Kd / Ke / Ks / Pr / Pm / Pc / Tr / bump)", scenes.build("T1", W, H))
