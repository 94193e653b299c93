"""Host-side scene DSL: Python face of libpt_host.so (include/pt_scene.h).

Mirrors the reference's `scene` class one method per method
(/root/reference/src/Main/dispatch.java:866-1062): addMaterial, setLastMtl, addObject, addTri,
addEllipsoid, addImplicit; `pack()` is BVH.allBVHtoList() + the SSBO packers (:270-329, :386-534).
Errors surface as RuntimeError, the way the reference throws RuntimeException.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

# SSBO binding points produced by the packers (frag.glsl:23-77)
BINDINGS = {3: np.float32, 5: np.float32, 7: np.float32, 10: np.float32, 11: np.int32, 12: np.int32, 13: np.int32, 14: np.float32}


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libpt_host.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        L = C.CDLL(path)
        L.pts_create.restype = C.c_void_p
        L.pts_destroy.argtypes = [C.c_void_p]
        L.pts_last_error.restype = C.c_char_p
        L.pts_add_material.argtypes = [C.c_void_p, C.c_char_p]
        L.pts_set_last_mtl.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_double), C.c_int]
        d3 = C.POINTER(C.c_double)
        L.pts_add_object.argtypes = [C.c_void_p, C.c_char_p, C.c_int, d3, d3, d3, C.c_char_p]
        L.pts_add_object_text.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_int, d3, d3, d3, C.c_char_p]
        L.pts_add_tri.argtypes = [C.c_void_p, d3, d3, d3, C.c_int]
        L.pts_add_ellipsoid.argtypes = [C.c_void_p, d3, d3, d3, C.c_float, C.c_int]
        L.pts_add_implicit.argtypes = [C.c_void_p, C.c_int, d3, d3, d3, C.c_int]
        L.pts_add_texture.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
        L.pts_texture_count.argtypes = [C.c_void_p]
        L.pts_texture_path.argtypes = [C.c_void_p, C.c_int]; L.pts_texture_path.restype = C.c_char_p
        L.pts_texture_name.argtypes = [C.c_void_p, C.c_int]; L.pts_texture_name.restype = C.c_char_p
        L.pts_parse_mtls.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
        L.pts_set_bvh_builder.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.pts_pack.argtypes = [C.c_void_p]
        L.pts_get_buffer.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        L.pts_count.argtypes = [C.c_void_p, C.c_int]
        L.pts_count.restype = C.c_int64
        _LIB = L
    return _LIB


def _d3(v):
    if np.isscalar(v):
        v = (v, v, v)
    return (C.c_double * 3)(*[float(x) for x in v])


class Scene:
    """The reference's static scene lists (dispatch.java:94-131) as one object."""

    def __init__(self):
        self._L = _lib()
        self._h = C.c_void_p(self._L.pts_create())

    def __del__(self):
        try:
            if self._h:
                self._L.pts_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def _check(self, rc):
        if rc < 0:
            raise RuntimeError(self._L.pts_last_error().decode())
        return rc

    def use_gpu_bvh_builder(self, device=0, enable=True):
        """Objects added from now on get their BVH from pt_build_bvh (libpt_hip.so, the same tree built on the GPU) instead of the
        recursive CPU builder.  Fails loudly when the HIP library is missing."""
        if not enable:
            self._check(self._L.pts_set_bvh_builder(self._h, None, 0, None))
            return
        from . import renderer
        hip = renderer.lib()
        self._check(self._L.pts_set_bvh_builder(self._h, C.cast(hip.pt_build_bvh, C.c_void_p), int(device), C.cast(hip.pt_last_error, C.c_void_p)))

    def addMaterial(self, name):
        return self._check(self._L.pts_add_material(self._h, name.encode()))

    def setLastMtl(self, prop, val):
        vals = [float(val)] if np.isscalar(val) else [float(x) for x in val]
        arr = (C.c_double * len(vals))(*vals)
        self._check(self._L.pts_set_last_mtl(self._h, prop.encode(), arr, len(vals)))

    def addTexture(self, path, name):
        """textures.add(path); textureNames.add(name) (dispatch.java:221-222: the sky is entry 0) -> index"""
        return self._check(self._L.pts_add_texture(self._h, path.encode(), name.encode()))

    def textures(self):
        """[(path, name)] of the texture table: entry i goes to pt_set_texture(i, ...) once decoded"""
        n = self._L.pts_texture_count(self._h)
        return [(self._L.pts_texture_path(self._h, i).decode(), self._L.pts_texture_name(self._h, i).decode()) for i in range(n)]

    def load_textures(self):
        """{index: (h, w, 4) uint8} decoded like stbi_load(path, 4 channels) does (dispatch.java:343): top row first"""
        from PIL import Image
        out = {}
        for i, (path, _) in enumerate(self.textures()):
            if path and os.path.exists(path):
                out[i] = np.asarray(Image.open(path).convert("RGBA"), dtype=np.uint8).copy()
        return out

    def parseMtls(self, filePath, parentDirectoryPath):
        pd = None if parentDirectoryPath is None else parentDirectoryPath.encode()
        self._check(self._L.pts_parse_mtls(self._h, filePath.encode(), pd))

    def addObject(self, filepath, material, scale=1.0, shift=0.0, rot=0.0, parentDirectory=None):
        pd = None if parentDirectory is None else parentDirectory.encode()
        self._check(self._L.pts_add_object(self._h, filepath.encode(), int(material), _d3(scale), _d3(shift), _d3(rot), pd))

    def addObjectText(self, obj_text, material, scale=1.0, shift=0.0, rot=0.0, parentDirectory=None):
        pd = None if parentDirectory is None else parentDirectory.encode()
        b = obj_text.encode() if isinstance(obj_text, str) else obj_text
        self._check(self._L.pts_add_object_text(self._h, b, len(b), int(material), _d3(scale), _d3(shift), _d3(rot), pd))

    def addTri(self, v1, v2, v3, m):
        self._check(self._L.pts_add_tri(self._h, _d3(v1), _d3(v2), _d3(v3), int(m)))

    def addEllipsoid(self, c, stretch, rot, radius, m):
        self._check(self._L.pts_add_ellipsoid(self._h, _d3(c), _d3(stretch), _d3(rot), float(radius), int(m)))

    def addImplicit(self, fn, shift, scale, rot, m):
        self._check(self._L.pts_add_implicit(self._h, int(fn), _d3(shift), _d3(scale), _d3(rot), int(m)))

    def count(self, what):
        names = {"triangles": 0, "nodes": 1, "objects": 2, "materials": 3, "ellipsoids": 4, "leaf_indices": 5, "max_depth": 6, "max_leaf": 7}
        return int(self._L.pts_count(self._h, names[what]))

    def pack(self):
        """-> {binding: numpy array} for bindings 3,5,7,10,11,12,13,14 (copies, like glBufferData)."""
        self._check(self._L.pts_pack(self._h))
        out = {}
        for b, dt in BINDINGS.items():
            p, n = C.c_void_p(), C.c_size_t()
            self._check(self._L.pts_get_buffer(self._h, b, C.byref(p), C.byref(n)))
            cnt = n.value // 4
            if cnt == 0:
                out[b] = np.zeros(0, dtype=dt)
            else:
                out[b] = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float if dt == np.float32 else C.c_int32)), shape=(cnt,)).copy()
        return out
