"""ctypes face of oracle/liboracle.so — TEST INFRASTRUCTURE (parity oracle), not product code.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
COUNTERS = ["segments", "nodes", "tritests", "hitupd", "samples", "boxtests", "rng", "ellip", "inner", "leftnext", "rightnext", "leafnext", "leafnext1", "xnodes", "xtris", "xdiff", "xpre"]


class _Scene(C.Structure):
    _fields_ = [("origin", C.c_void_p), ("rotation", C.c_void_p), ("mouse", C.c_void_p), ("tris", C.c_void_p), ("n_tris", C.c_int64),
                ("params", C.c_void_p), ("imp", C.c_void_p), ("ellip", C.c_void_p), ("bvhdata", C.c_void_p), ("bvhtree", C.c_void_p),
                ("n_nodes", C.c_int64), ("leaf_tris", C.c_void_p), ("n_leaf_tris", C.c_int64), ("obj_indices", C.c_void_p), ("mtl", C.c_void_p),
                ("n_mtl_floats", C.c_int64), ("sky", C.c_void_p), ("sky_w", C.c_int32), ("sky_h", C.c_int32),
                ("tex", C.c_void_p), ("tex_w", C.c_void_p), ("tex_h", C.c_void_p), ("n_tex", C.c_int32)]


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.orc_render.argtypes = [C.POINTER(_Scene), C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.orc_autofocus.argtypes = [C.POINTER(_Scene)]
        L.orc_autofocus.restype = C.c_float
        L.orc_ray_scene.argtypes = [C.POINTER(_Scene), C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_rng.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_math.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
        L.orc_rotate.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.orc_set_whatif.argtypes = [C.c_int, C.c_float]
        L.orc_display.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        if not L.orc_has_fma():
            raise RuntimeError("oracle needs a CPU with FMA (built with -mfma)")
        _LIB = L
    return _LIB


class Scene:
    """Holds the SSBO contents (dict binding -> array) + texture 0 for the oracle."""

    def __init__(self, buffers, sky, textures=None):
        """textures: optional {index: (h, w, 4) uint8} for the bindless table beyond the sky (index 0)"""
        f32 = lambda b: np.ascontiguousarray(buffers[b], dtype=np.float32)
        i32 = lambda b: np.ascontiguousarray(buffers[b], dtype=np.int32)
        self.keep = dict(origin=f32(0), rotation=f32(1), mouse=f32(2), tris=f32(3), params=f32(4), imp=f32(5), ellip=f32(7), bvhdata=f32(10),
                         bvhtree=i32(11), leaf=i32(12), obj=i32(13), mtl=f32(14), sky=np.ascontiguousarray(sky, dtype=np.uint8))
        k = self.keep
        if k["tris"].size == 0:
            k["tris"] = np.zeros(40, np.float32)
        for name in ("bvhdata", "bvhtree", "leaf"):
            if k[name].size == 0:
                k[name] = np.zeros(8, k[name].dtype)
        p = lambda a: a.ctypes.data
        s = _Scene()
        s.origin, s.rotation, s.mouse, s.tris, s.n_tris = p(k["origin"]), p(k["rotation"]), p(k["mouse"]), p(k["tris"]), buffers[3].size // 40
        s.params, s.imp, s.ellip, s.bvhdata, s.bvhtree = p(k["params"]), p(k["imp"]), p(k["ellip"]), p(k["bvhdata"]), p(k["bvhtree"])
        s.n_nodes, s.leaf_tris, s.n_leaf_tris, s.obj_indices = buffers[11].size // 3, p(k["leaf"]), buffers[12].size, p(k["obj"])
        s.mtl, s.n_mtl_floats, s.sky, s.sky_w, s.sky_h = p(k["mtl"]), k["mtl"].size, p(k["sky"]), k["sky"].shape[1], k["sky"].shape[0]
        tex = {0: k["sky"]}
        for idx, arr in (textures or {}).items():
            if idx != 0:
                tex[int(idx)] = np.ascontiguousarray(arr, dtype=np.uint8)
        n = max(tex) + 1
        self.keep["tex"] = tex
        self.keep["tex_ptr"] = (C.c_void_p * n)(*[tex[i].ctypes.data if i in tex else None for i in range(n)])
        self.keep["tex_w"] = np.array([tex[i].shape[1] if i in tex else 0 for i in range(n)], np.int32)
        self.keep["tex_h"] = np.array([tex[i].shape[0] if i in tex else 0 for i in range(n)], np.int32)
        s.tex = C.cast(self.keep["tex_ptr"], C.c_void_p); s.tex_w = p(self.keep["tex_w"]); s.tex_h = p(self.keep["tex_h"]); s.n_tex = n
        self.c = s

    @classmethod
    def from_workload(cls, wl):
        return cls(wl.buffers, wl.sky, getattr(wl, "textures", None))


def render(scene, W, H, frame_count, seed, frame=None, nthreads=1, x0=0, xs=1, y0=0, ys=1, counters=None):
    """One frame (one glDrawArrays). `frame` (H,W,4) f32 is updated in place (created zeroed if None)."""
    if frame is None:
        frame = np.zeros((H, W, 4), dtype=np.float32)
    assert frame.dtype == np.float32 and frame.flags.c_contiguous and frame.shape == (H, W, 4)
    cnt = np.zeros(len(COUNTERS), dtype=np.uint64) if counters is None else counters
    rc = lib().orc_render(C.byref(scene.c), W, H, frame_count, seed, frame.ctypes.data, nthreads, x0, xs, y0, ys, cnt.ctypes.data)
    if rc:
        raise RuntimeError(f"oracle: orc_render failed ({rc})")
    return frame, cnt


def render_frames(scene, W, H, first_frame, n_frames, seeds, frame=None, nthreads=1, **kw):
    cnt = np.zeros(len(COUNTERS), dtype=np.uint64)
    for i in range(n_frames):
        frame, _ = render(scene, W, H, first_frame + i, int(seeds[i]), frame, nthreads, counters=cnt, **kw)
    return frame, cnt


def rng(state, n):
    st = np.array([state], dtype=np.uint32)
    res, rnd = np.zeros(n, np.uint32), np.zeros(n, np.float32)
    lib().orc_rng(st.ctypes.data, n, res.ctypes.data, rnd.ctypes.data)
    return int(st[0]), res, rnd


def math(fn, x, y=None):
    names = {"sin": 0, "cos": 1, "log": 2, "exp": 3, "atan2": 4, "asin": 5}
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.zeros_like(x) if y is None else np.ascontiguousarray(y, dtype=np.float32)
    out = np.empty_like(x)
    lib().orc_math(names[fn], x.ctypes.data, y.ctypes.data, out.ctypes.data, x.size)
    return out


def ray_scene(scene, o, d):
    o, d, out = np.array(o, np.float32), np.array(d, np.float32), np.zeros(8, np.float32)
    code = lib().orc_ray_scene(C.byref(scene.c), o.ctypes.data, d.ctypes.data, out.ctypes.data)
    return code, out


def autofocus(scene):
    return float(lib().orc_autofocus(C.byref(scene.c)))


def rotate(p, rot, back=False):
    p, rot, out = np.array(p, np.float32), np.array(rot, np.float32), np.zeros(3, np.float32)
    lib().orc_rotate(p.ctypes.data, rot.ctypes.data, 1 if back else 0, out.ctypes.data)
    return out


def display(frame, frame_count, java_bytes=True):
    """8-bit screenshot image (H, W, 3) uint8, top row first, of a FRAME accumulator."""
    H, W = frame.shape[:2]
    f = np.ascontiguousarray(frame, dtype=np.float32)
    out = np.zeros((H, W, 3), dtype=np.uint8)
    lib().orc_display(f.ctypes.data, W, H, int(frame_count), 1 if java_bytes else 0, out.ctypes.data)
    return out


def set_whatif(mode, margin=1.0 / 64):
    """statistics only: count what another order of rayScene's object loop would visit on the same rays (frag_oracle.cpp: xObjectLoop); 0 = off"""
    lib().orc_set_whatif(int(mode), float(margin))
