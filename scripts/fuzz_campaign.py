#!/usr/bin/env python3
"""A longer run of tests/test_gpu_parity.py's randomized scenes than the suite holds: fuzz_campaign.py <first seed> <count>

Each seed: a random scene (meshes with smooth / flat normals, with and without vt, ellipsoids, mapped materials, random parameter blocks),
rendered through the C ABI with one of the intersect kernel's block sizes and stack widths, as one stream and as a two-stream group, both modes
(RAYTRACING 1 / 0), compared bit for bit (frames and traversal counters) with the oracle.  Prints one line per seed; exit code = mismatches.
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np  # noqa: E402
import ptimport  # noqa: E402

pt = ptimport.load()
from pathtracer_0_amd import build, renderer  # noqa: E402
build.build_host()
import oracle  # noqa: E402
oracle.lib()
import test_gpu_parity as T  # noqa: E402

first, count = int(sys.argv[1]), int(sys.argv[2])
bad = 0
t0 = time.time()
for seed in range(first, first + count):
    wl = T._random_workload(pt, seed, ellipsoid_maps=seed % 3 == 0 and seed % 4 != 3, many_groups=seed % 4 == 3)
    opts = dict(extend_tpb=[256, 64, 256, 1024, 256][seed % 5], stack_mode=[-1, -1, 1, 2][seed % 4], refill_min=[1, 8, 24, 48][(seed // 5) % 4],
                index_stack_8bit=(seed // 2) % 3, asm_node_layout=[-1, 0, 1][seed % 3], asm_tpb=[0, 256, 1024][(seed // 3) % 3], asm_loop=[-1, 0, 1][(seed // 7) % 3], asm_root_cull=(seed // 5) % 2)
    try:
        got, ref, cnt, ocnt = T.render_both(pt, oracle, renderer, wl, 3, **opts)
        T.assert_same(got, ref, cnt, ocnt)
        d = wl.with_params(RAYTRACING=0)
        got, ref, cnt, ocnt = T.render_both(pt, oracle, renderer, d, 2, **opts)
        T.assert_same(got, ref, cnt, ocnt)
        # the two-stream group (the production form), overlapped batches
        seeds = T.seeds_for(pt, 1, 4)
        r = renderer.Renderer(wl.W, wl.H, devices=[0, 0])
        for k, v in opts.items():
            r.set_option(k, v)
        r.load_workload(wl); r.reset_frame()
        r.render_batch_async(1, seeds[:2]); r.render_batch_async(3, seeds[2:])
        got = r.read_frame(); r.close()
        ref, _ = oracle.render_frames(oracle.Scene.from_workload(wl), wl.W, wl.H, 1, 4, seeds, nthreads=8)
        T.assert_same(got, ref)
        print(f"seed {seed} ok {opts} ({time.time() - t0:.0f} s)", flush=True)
    except AssertionError as e:
        bad += 1
        print(f"seed {seed} MISMATCH {opts}: {str(e)[:300]}", flush=True)
print(f"{count} seeds, {bad} mismatches")
sys.exit(1 if bad else 0)
