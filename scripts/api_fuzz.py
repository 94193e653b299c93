#!/usr/bin/env python3
"""Random sequences of the render path's entry points against a model built from the oracle:  api_fuzz.py <first seed> <count>

Per seed: a small scene, a one-stream or two-stream context, a random path-pool size, then ~14 random calls out of
  pt_render / pt_render_batch / pt_render_batch_async (frame counters running on, or restarting at 1: frag.glsl:924-933 stores instead of adding),
  bursts of 6-40 one-frame pt_render_batch_async calls (the reference's loop), pt_write_frame of one of the ring's images,
  pt_next_image, pt_reset_frame, pt_read_frame, pt_gather_image of an older image, uploads of ORIGIN / ROTATION / Parameters between batches
  (SAMPLE_RES, MAX_BOUNCES, RAYTRACING, AUTO_FOCUS), pt_set_option(path_slots).
The model keeps the ring of four FRAME images as numpy arrays and renders every submitted frame with the oracle and the inputs current at its
submission; every image read is compared bit for bit.  Exit code = mismatching seeds.
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np  # noqa: E402
import torch  # noqa: E402
import ptimport  # noqa: E402

pt = ptimport.load()
from pathtracer_0_amd import build, renderer, scenes, shard  # noqa: E402
build.build_host()
import oracle  # noqa: E402
oracle.lib()


def same(a, b):
    return bool(((a == b) | (np.isnan(a) & np.isnan(b))).all())


def one(seed, verbose=False):
    rs = np.random.RandomState(seed)
    name = str(rs.choice(["C1", "C2", "C3", "T1"]))
    W, H = [(64, 36), (48, 27), (80, 48)][rs.randint(3)]
    wl = scenes.build(name, W, H)
    bufs = {k: np.array(v, copy=True) for k, v in wl.buffers.items()}
    two = bool(rs.randint(2))
    r = renderer.Renderer(W, H, devices=[0, 0]) if two else renderer.Renderer(W, H)
    r.set_option("path_slots", int(rs.choice([0, 0, 1536, 4096, 65536])))
    r.load_workload(wl)
    r.reset_frame()
    scene = oracle.Scene(bufs, wl.sky, wl.textures)
    ring = [np.zeros((H, W, 4), np.float32)]            # ring[-1] = current image, ring[-1 - age] = `age` pt_next_image calls ago
    fc = 1

    class _Log(list):                                   # API_FUZZ_TRACE=<file>: every call is written out BEFORE it is made (a native abort leaves the last one behind)
        def append(self, x):
            list.append(self, x)
            if os.environ.get("API_FUZZ_TRACE"):
                with open(os.environ["API_FUZZ_TRACE"], "a") as f:
                    f.write(f"seed {seed}: {x}\n")
    log = _Log()

    def submit(kind, n):
        nonlocal fc
        if rs.rand() < 0.12:
            fc = 1                                      # u_frameCount == 1: store, do not add (frag.glsl:926-928)
        seeds = [int(rs.randint(0, 10000)) for _ in range(n)]
        log.append(f"{kind}(first={fc}, n={n})")
        if kind == "render":
            r.render(fc, seeds[0])
        elif kind == "batch":
            r.render_batch(fc, seeds)
        else:
            r.render_batch_async(fc, seeds)
        if os.environ.get("API_FUZZ_SELFTEST") == "1" and kind == "async" and n > 2:
            seeds = seeds[:-1] + [seeds[-1] ^ 1]        # self-test of the checker: a wrong seed in the model must be noticed
        oracle.render_frames(scene, W, H, fc, n, seeds, frame=ring[-1], nthreads=8)
        fc += n

    def check(age):
        log.append(f"check(age={age})")
        if age == 0 and rs.rand() < 0.5:
            got = r.read_frame()
        else:
            t = torch.as_tensor(shard._DevArray(r.gather_image(age), (H, W, 4)), device=torch.device("cuda", 0))
            r.stream_wait()
            got = t.cpu().numpy()
        return same(got, ring[-1 - age])

    ok = True
    for _ in range(int(rs.randint(8, 20))):
        op = rs.choice(["render", "batch", "async", "async", "burst", "next", "reset", "check", "check_old", "origin", "params", "slots", "write"])
        if op in ("render", "batch", "async"):
            submit(op, 1 if op == "render" else int(rs.randint(1, 5)))
        elif op == "burst" and os.environ.get("API_FUZZ_NO_BURST") != "1":      # the reference's loop: one draw per call, many of them, nothing waited for
            for _ in range(int(rs.randint(6, 40))):
                submit("async", 1)
        elif op == "write" and hasattr(renderer.lib(), "pt_write_frame") and os.environ.get("API_FUZZ_NO_WRITE") != "1":      # pt_write_frame: one of the ring's images (the model's copy) becomes the current accumulator
            age = int(rs.randint(0, len(ring)))
            log.append(f"write_frame(image of age {age})")
            src = ring[-1 - age].copy()
            r.write_frame(src); ring[-1][:] = src
            fc = int(src[0, 0, 3]) + 1 if src[0, 0, 3] >= 1 else 1
        elif op == "next":
            log.append("next_image")
            r.next_image(); ring.append(np.zeros((H, W, 4), np.float32)); ring[:] = ring[-4:]; fc = 1
        elif op == "reset":
            log.append("reset_frame")
            r.reset_frame(); ring[-1][:] = 0; fc = 1
        elif op == "check":
            ok &= check(0)
        elif op == "check_old" and len(ring) > 1:
            ok &= check(int(rs.randint(1, len(ring))))
        elif op == "origin":
            b = 0 if rs.rand() < 0.6 else 1
            bufs[b] = (bufs[b] + rs.uniform(-0.05, 0.05, 3).astype(np.float32)).astype(np.float32)
            log.append(f"set_buffer({b})")
            r.set_buffer(b, bufs[b]); scene = oracle.Scene(bufs, wl.sky, wl.textures)
        elif op == "params":
            p = bufs[4].copy()
            which = rs.randint(4)
            if which == 0: p[4] = float(rs.choice([1, 2, 4, 8]))
            elif which == 1: p[5] = float(rs.choice([1, 2, 4, 8]))
            elif which == 2: p[9] = 0.0 if p[9] == 1.0 else 1.0
            else: p[11] = 0.0 if p[11] == 1.0 else 1.0
            bufs[4] = p
            log.append(f"set_buffer(4: SAMPLE_RES {p[4]} MAX_BOUNCES {p[5]} RAYTRACING {p[9]} AUTO_FOCUS {p[11]})")
            r.set_buffer(4, p); scene = oracle.Scene(bufs, wl.sky, wl.textures)
        elif op == "slots":
            v = int(rs.choice([0, 1024, 2048, 8192]))
            log.append(f"path_slots={v}")
            r.set_option("path_slots", v)
        if not ok:
            break
    if ok:
        for age in range(len(ring)):
            ok &= check(age)
    r.close()
    if not ok or verbose:
        print(f"seed {seed} {'ok' if ok else 'MISMATCH'} {name} {W}x{H} {'two streams' if two else 'one stream'}: " + "; ".join(log), flush=True)
    return ok


if __name__ == "__main__":
    first, count = int(sys.argv[1]), int(sys.argv[2])
    t0 = time.time()
    bad = 0
    for s in range(first, first + count):
        try:
            bad += 0 if one(s, verbose=os.environ.get("API_FUZZ_VERBOSE") == "1") else 1
        except Exception as e:      # an error code from the library is a finding too
            bad += 1
            print(f"seed {s} ERROR {type(e).__name__}: {str(e)[:300]}", flush=True)
        if (s - first) % 50 == 49:
            print(f"... {s - first + 1} seeds, {bad} bad, {time.time() - t0:.0f} s", flush=True)
    print(f"{count} sequences, {bad} mismatches / errors")
    sys.exit(1 if bad else 0)
