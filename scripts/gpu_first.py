import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np
import ptimport; pt = ptimport.load()
from pathtracer_0_amd import renderer
import oracle
W, H, F = 96, 54, 2
for name in ("C2", "C3", "C1"):
    wl = pt.scenes.build(name, W, H)
    seeds = [pt.scenes.frame_seed(f) for f in range(1, F + 1)]
    r = renderer.Renderer(W, H)
    r.set_option("count_stats", 1)
    r.load_workload(wl); r.reset_frame(); r.reset_counters()
    t = time.time(); r.render_batch(1, seeds); got = r.read_frame(); dt = time.time() - t
    cnt = r.counters(); r.close()
    sc = oracle.Scene.from_workload(wl)
    ref, ocnt = oracle.render_frames(sc, W, H, 1, F, seeds, nthreads=8)
    neq = (~((got == ref) | (np.isnan(got) & np.isnan(ref)))).sum()
    print(name, "differing floats:", int(neq), "of", got.size, "maxabs", float(np.nanmax(np.abs(got - ref))), "time %.3f" % dt)
    print("  gpu", cnt); print("  orc", dict(zip(oracle.COUNTERS, ocnt.tolist())))
