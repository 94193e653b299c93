"""Oracle statistics (CPU, test infrastructure): what another order of rayScene's object loop (frag.glsl:563-577) would visit on the SAME rays of a
workload, and how many hit records would differ from the reference's.  python scripts/object_order_whatif.py [C3 C4 C5 C6] [--step 8]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import ptimport  # noqa
from oracle import oracle
pt = ptimport.load()
scenes = pt.scenes

MODES = {0: "reference (index order)", 1: "nearest first, <= below the winner", 2: "nearest first, <= and margin", 3: "pre: nearest BVH; then index order from U",
         4: "pre: nearest first (all); then index order from U", 5: "pre: nearest first until a hit; then from U", 6: "pre: nearest first until the first triangle hit; then from U",
         7: "the nearest root box first, the rest in index order", 8: "the two nearest first, the rest in index order", 9: "index order over the root boxes the ray meets (today's kernel)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("configs", nargs="*", default=["C3", "C4", "C5", "C6"])
    ap.add_argument("--step", type=int, default=8)
    ap.add_argument("--frames", type=int, default=1)
    ap.add_argument("--modes", default="1,2,3,4,5,6")
    ap.add_argument("--margin", type=float, default=1.0 / 64)
    a = ap.parse_args()
    for name in a.configs:
        wl = scenes.build(name)
        sc = oracle.Scene.from_workload(wl)
        for mode in [int(m) for m in a.modes.split(",")]:
            oracle.set_whatif(mode, a.margin)
            cnt = np.zeros(len(oracle.COUNTERS), np.uint64)
            for f in range(a.frames):
                oracle.render(sc, wl.W, wl.H, f + 1, scenes.frame_seed(f), None, nthreads=os.cpu_count(), x0=0, xs=a.step, y0=0, ys=a.step, counters=cnt)
            c = dict(zip(oracle.COUNTERS, [int(v) for v in cnt]))
            S = c["segments"]
            print(f"{name} mode {mode} [{MODES[mode]}]: segments {S}  reference nodes/seg {c['nodes']/S:.2f} tris/seg {c['tritests']/S:.2f}  |  what-if nodes/seg {c['xnodes']/S:.2f}"
                  f" (pre-pass {c['xpre']/S:.2f}) tris/seg {c['xtris']/S:.2f}  hit records that differ {c['xdiff']} of {S}", flush=True)
        oracle.set_whatif(0)


if __name__ == "__main__":
    main()
