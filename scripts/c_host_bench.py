#!/usr/bin/env python3
"""bench.py's schedule from a plain C host (tests/c/bench_client.c): the number a Java / C caller of libpt_hip.so sees — ROCm's own HIP runtime, no torch in the
process — beside bench.py's (review item 6).  This script only dumps the workload's SSBO contents, builds the client and starts it; it loads neither torch nor
the library.   usage: c_host_bench.py [--config C3] [--steps 20] [--warmup 5] [--streams 2] [--frames-per-step 32] [--width W --height H]"""
import argparse
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

import ptimport

pt = ptimport.load()
from pathtracer_0_amd import build, scenes  # noqa: E402


def build_client(out_dir):
    exe = os.path.join(out_dir, "bench_client")
    lib = os.path.join(ROOT, "pathtracer-0_amd")
    build.build_hip()
    subprocess.check_call(["gcc", "-std=c99", "-O2", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"), "-o", exe, os.path.join(ROOT, "tests", "c", "bench_client.c"),
                           "-L" + lib, "-lpt_hip", "-ldl", "-Wl,-rpath," + lib])
    return exe


def dump_workload(wl, d):
    for b in (0, 1, 2, 3, 4, 5, 7, 10, 11, 12, 13, 14):
        np.ascontiguousarray(wl.buffers[b]).tofile(os.path.join(d, f"binding_{b}.bin"))
    sky = np.ascontiguousarray(wl.sky, dtype=np.uint8)
    sky.tofile(os.path.join(d, "sky.bin"))
    return sky.shape[1], sky.shape[0]


def run(config="C3", steps=20, warmup=5, streams=2, fps=None, W=None, H=None, workdir=None):
    cfg = scenes.CONFIGS[config]
    wl = scenes.build(config, W, H)
    fps = fps or cfg["spp"] // cfg["sample_res"]
    d = workdir or tempfile.mkdtemp(prefix="pt_c_host_")
    skyW, skyH = dump_workload(wl, d)
    exe = build_client(d)
    env = dict(os.environ); env.setdefault("GPU_MAX_HW_QUEUES", "8")
    out = subprocess.run([exe, d, str(wl.W), str(wl.H), str(skyW), str(skyH), str(wl.sample_res), str(fps), str(steps), str(warmup), str(streams)], capture_output=True, text=True, timeout=900, env=env)
    if out.returncode != 0:
        raise RuntimeError(out.stdout + out.stderr)
    return [l for l in out.stdout.splitlines() if l.startswith("{")][-1]


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C3"); ap.add_argument("--steps", type=int, default=20); ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--streams", type=int, default=2); ap.add_argument("--frames-per-step", type=int, default=None)
    ap.add_argument("--width", type=int, default=None); ap.add_argument("--height", type=int, default=None)
    a = ap.parse_args()
    print(run(a.config, a.steps, a.warmup, a.streams, a.frames_per_step, a.width, a.height))
