#!/usr/bin/env python3
"""Generates tests/golden/*.npz.

The reference ships no golden vectors and cannot be executed here, so these fixtures are produced by the
parity oracle (oracle/frag_oracle.cpp) through the host-side scene producers; they freeze the numeric contract and the
scene packers so that a later change to either shows up as a diff, on the CPU suite and on the GPU suite alike.
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ptimport  # noqa: E402

pt = ptimport.load()
import oracle  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = {"C1": (48, 48, 2, {}), "C2": (48, 27, 2, {}), "C3": (48, 27, 3, {}), "C5": (48, 27, 2, dict(subdiv=2)),
         "T1": (48, 27, 2, {}),                       # material texture maps: the texture table travels as tex<i>
         "C5direct": (48, 27, 2, dict(subdiv=2))}     # RAYTRACING = 0 (directDiffuse, N2) on the subsurface / clearcoat scene


def main():
    for name, (W, H, frames, kw) in CASES.items():
        wl = pt.scenes.build(name.replace("direct", ""), W, H, **kw)
        if name.endswith("direct"):
            wl = wl.with_params(RAYTRACING=0)
        seeds = [pt.scenes.frame_seed(f) for f in range(1, frames + 1)]
        frame, cnt = oracle.render_frames(oracle.Scene.from_workload(wl), W, H, 1, frames, seeds, nthreads=4)
        out = {f"b{k}": v for k, v in wl.buffers.items()}
        out.update({f"tex{i}": a for i, a in wl.textures.items()})
        np.savez_compressed(os.path.join(HERE, f"{name}_{W}x{H}_{frames}f.npz"), frame=frame, counters=cnt, seeds=np.array(seeds, np.int32), sky=wl.sky, **out)
        print(name, W, H, frames, dict(zip(oracle.COUNTERS, cnt.tolist())))
    # RNG known-answer vectors (frag.glsl:686-694) and math-contract samples
    st, rows = 0, []
    for start in (0, 1, 12345, 2083598, 0xFFFFFFFF):
        s2, res, rnd = oracle.rng(start, 8)
        rows.append((start, res, rnd.view(np.uint32)))
    x = np.linspace(-7, 7, 257, dtype=np.float32)
    u = ((np.arange(257, dtype=np.float64) * 16777259.0) % 2 ** 32 / 2 ** 32).astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "contract.npz"), rng_start=np.array([r[0] for r in rows], np.uint32), rng_result=np.stack([r[1] for r in rows]),
                        rng_random_bits=np.stack([r[2] for r in rows]), x=x, u=u, sin=oracle.math("sin", x), cos=oracle.math("cos", x), exp=oracle.math("exp", x * 10),
                        log=oracle.math("log", u), asin=oracle.math("asin", x / 7), atan2=oracle.math("atan2", x, x[::-1].copy()))


if __name__ == "__main__":
    main()
