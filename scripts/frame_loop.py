#!/usr/bin/env python3
"""The reference's own usage pattern: ONE frame (SAMPLE_RES spp) per draw call, camera standing still (dispatch.java:693-705).
Compares pt_render per frame (drains the GPU every frame, like glFinish after every draw) with pt_render_batch_async per frame
(the draw calls stay in flight, as the GL driver leaves them) and with one 32-frame batch."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ptimport  # noqa: E402

pt = ptimport.load()
from pathtracer_0_amd import renderer, scenes  # noqa: E402

W, H, N = 1920, 1080, int(os.environ.get("FRAMES", "64"))
wl = scenes.build("C3", W, H)
# usage: frame_loop.py [streams] [path_slots]   (streams > 1: pt_create_multi with GPU 0 listed that many times)
streams = int(sys.argv[1]) if len(sys.argv) > 1 else 1
r = renderer.Renderer(W, H, devices=[0] * streams) if streams > 1 else renderer.Renderer(W, H)
if len(sys.argv) > 2 and int(sys.argv[2]):
    r.set_option("path_slots", int(sys.argv[2]))
print(f"streams {streams}, path slots {sys.argv[2] if len(sys.argv) > 2 else 'automatic'}")
r.load_workload(wl)
seeds = [scenes.frame_seed(f) for f in range(1, N + 1)]


def timed(fn):
    r.reset_frame(); r.render(1, seeds[0]); r.synchronize(); r.reset_frame()      # warm
    t = time.perf_counter(); fn(); r.synchronize(); dt = time.perf_counter() - t
    return dt, r.read_frame().copy()


def per_frame_sync():
    for f in range(N):
        r.render(f + 1, seeds[f])


def per_frame_async():
    for f in range(N):
        r.render_batch_async(f + 1, seeds[f:f + 1])


def one_batch():
    for k in range(0, N, 32):
        r.render_batch(k + 1, seeds[k:k + 32])


def batches_async():
    for k in range(0, N, 32):
        r.render_batch_async(k + 1, seeds[k:k + 32])


res = {}
for name, fn in (("pt_render per frame", per_frame_sync), ("pt_render_batch_async per frame", per_frame_async), ("32-frame batches (synchronous)", one_batch), ("32-frame batches (asynchronous)", batches_async)):
    dt, img = timed(fn)
    res[name] = img
    print(f"{name:34s} {N} frames x 8 spp at {W}x{H}: {dt * 1e3 / N:7.2f} ms/frame  {W * H * 8 * N / dt / 1e6:8.1f} Msamples/s", flush=True)
imgs = list(res.values())
print("all images bit-identical:", all((imgs[0] == im).all() for im in imgs[1:]))
r.close()
