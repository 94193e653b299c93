#!/usr/bin/env python3
"""Start-up cost of the BVH build: CPU mirror of the Java builder vs pt_build_bvh on the GPU, same tree.

usage: bvh_bench.py [quads_x quads_y]...     each pair = one height-field mesh of 2*qx*qy triangles in ONE object
Prints per mesh: OBJ parse + build through libpt_host.so with either builder, and the bare pt_build_bvh call.
"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ptimport  # noqa: E402

pt = ptimport.load()
from pathtracer_0_amd import hostlib, renderer  # noqa: E402


def heightfield(qx, qy, seed=3):
    rs = np.random.RandomState(seed)
    x, y = np.meshgrid(np.linspace(-1, 1, qx + 1), np.linspace(-1, 1, qy + 1), indexing="ij")
    z = 0.15 * np.sin(7 * x) * np.cos(5 * y) + 0.01 * rs.rand(qx + 1, qy + 1)
    v = np.stack([x, z, y], -1).reshape(-1, 3)
    i = (np.arange(qx)[:, None] * (qy + 1) + np.arange(qy)[None, :]).reshape(-1)
    f = np.concatenate([np.stack([i, i + 1, i + qy + 2], -1), np.stack([i, i + qy + 2, i + qy + 1], -1)]) + 1
    return v, f


def obj_text(v, f):
    vl = np.char.add(np.char.add(np.char.add("v ", np.char.mod("%.9g", v[:, 0])), np.char.add(" ", np.char.mod("%.9g", v[:, 1]))), np.char.add(" ", np.char.mod("%.9g", v[:, 2])))
    fs = np.char.mod("%d", f)
    fl = np.char.add(np.char.add(np.char.add("f ", np.char.add(fs[:, 0], "//1 ")), np.char.add(fs[:, 1], "//1 ")), np.char.add(fs[:, 2], "//1"))
    return "o mesh\nvn 0 1 0\n" + "\n".join(vl.tolist()) + "\n" + "\n".join(fl.tolist()) + "\n"


def main(argv):
    sizes = [(int(argv[k]), int(argv[k + 1])) for k in range(0, len(argv) - 1, 2)] or [(224, 224), (708, 708)]
    hip = renderer.lib()
    hip.pt_build_bvh.argtypes = [C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    warm = hostlib.Scene(); warm.addMaterial("m"); warm.use_gpu_bvh_builder(0)      # HIP runtime start-up outside the timings
    warm.addObjectText("o warm\nvn 0 1 0\nv 0 0 0\nv 1 0 0\nv 0 0 1\nv 1 0 1\nf 1//1 2//1 3//1\nf 2//1 4//1 3//1\n", 0)
    for qx, qy in sizes:
        v, f = heightfield(qx, qy)
        n = len(f)
        text = obj_text(v, f).encode()
        res = {}
        packs = []
        for name in ("cpu", "gpu"):
            sc = hostlib.Scene(); sc.addMaterial("m")
            if name == "gpu":
                sc.use_gpu_bvh_builder(0)
            t = time.perf_counter(); sc.addObjectText(text, 0); res[name] = time.perf_counter() - t
            packs.append(sc.pack())
        same = all(np.array_equal(packs[0][b].view(np.uint32), packs[1][b].view(np.uint32)) for b in (3, 10, 11, 12, 13))
        # the bare builder call on the same triangles
        tv = v[f - 1]                                              # (n, 3, 3)
        tri9 = np.ascontiguousarray(np.concatenate([tv.min(1), tv.max(1), (tv[:, 0] + (tv[:, 1] + tv[:, 2])) / 3.0], 1))
        nn = C.c_int32(); dep = C.c_int32()
        bounds = np.zeros((2 * n, 6)); links = np.zeros((2 * n, 2), np.int32); leaf = np.zeros((2 * n, 2), np.int32); order = np.zeros(n, np.int32)
        t = time.perf_counter()
        rc = hip.pt_build_bvh(0, tri9.ctypes.data, n, C.byref(nn), bounds.ctypes.data, links.ctypes.data, leaf.ctypes.data, order.ctypes.data, C.byref(dep))
        tb = time.perf_counter() - t
        print(f"{n} triangles: parse+build CPU {res['cpu']:.3f} s | parse+build GPU {res['gpu']:.3f} s | pt_build_bvh alone {tb:.3f} s (rc {rc}, {nn.value} nodes, depth {dep.value}) | "
              f"=> CPU build ~{res['cpu'] - res['gpu'] + tb:.3f} s vs GPU {tb:.3f} s; packed buffers 3/10/11/12/13 bit-identical: {same}", flush=True)


if __name__ == "__main__":
    main(sys.argv[1:])
