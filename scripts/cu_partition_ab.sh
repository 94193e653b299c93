#!/bin/bash
# round 6 (c): spatial partition A/B — the intersect launches on CU-masked streams holding e eighths of every XCD's CUs, the shading launches on the complement
R=${GRAFT_REPO_ROOT:-$(pwd)}; T=${1:-r06c}; O=$R/gpurun_out/$T; mkdir -p $O; cd $R
# parity first: whole renders with the partition on (one and two streams) against the oracle
timeout -k 10 300 python3 - > $O/parity.txt 2>&1 <<'PY'
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "oracle"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np, torch, ptimport
pt = ptimport.load()
from pathtracer_0_amd import renderer, scenes
import oracle
for name, W, H in (("C3", 256, 144), ("C6", 192, 108)):
    wl = scenes.build(name, W, H); seeds = [scenes.frame_seed(f) for f in range(1, 4)]
    ref, _ = oracle.render_frames(oracle.Scene.from_workload(wl), W, H, 1, 3, seeds, nthreads=8)
    for devs in (None, [0, 0]):
        for e in (0, 6, 4):
            r = renderer.Renderer(W, H, devices=devs) if devs else renderer.Renderer(W, H)
            r.set_option("cu_partition", e); r.load_workload(wl); r.reset_frame(); r.render_batch(1, seeds[:1]); r.render_batch_async(2, seeds[1:]); got = r.read_frame().copy(); r.close()
            same = np.array_equal(got.view(np.uint32), ref.view(np.uint32)) or np.array_equal(got, ref, equal_nan=True)
            print(name, "streams", 2 if devs else 1, "cu_partition", e, "bit-identical to the oracle:", same, flush=True)
            assert same
print("PARITY_OK")
PY
tail -3 $O/parity.txt
grep -q PARITY_OK $O/parity.txt || exit 1
bash scripts/flag_ab.sh -r 2 -c "C3 C4" -t -- "" "--cu-partition 6" "--cu-partition 5" "--cu-partition 7" "--cu-partition 4" 2>&1 | tee $O/ab.txt
bash scripts/flag_ab.sh -r 1 -c "C3" -t -- "--cu-partition 6 --extend-blocks-per-cu 6" "--cu-partition 6 --streams 3" "--cu-partition 5 --streams 3" "--streams 1" "--streams 1 --cu-partition 6" 2>&1 | tee -a $O/ab.txt
