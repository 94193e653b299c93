"""Synthetic workloads C1..C5 of BASELINE.json / SURVEY.md §8(d).

No asset of the reference is usable (all live under C:\\Graphics\\... off-repo,
/root/reference/src/Main/dispatch.java:221-257), so every workload is generated here,
deterministically, as OBJ text + scene-DSL calls and pushed through the same host-side producers a
reference user would call (hostlib.Scene == the reference's `scene` class).  All meshes carry
explicit `vn`, every OBJ starts with an `o` line (Q-14) and every `o` group has
at least two separable triangles (Q-16).
"""
import math

import numpy as np

from . import hostlib

# (W, H, spp, bounces) per BASELINE.json config
CONFIGS = {
    "C1": dict(W=256, H=256, spp=4, bounces=4, sample_res=4),
    "C2": dict(W=1280, H=720, spp=64, bounces=8, sample_res=8),
    "C3": dict(W=1920, H=1080, spp=256, bounces=8, sample_res=8),
    "C4": dict(W=1920, H=1080, spp=1024, bounces=8, sample_res=8),
    "C5": dict(W=3840, H=2160, spp=4096, bounces=16, sample_res=8),
    # not a BASELINE config: the shape of the reference author's own scenes (multi-group OBJs + addEllipsoid, dispatch.java:245-264)
    "C6": dict(W=1920, H=1080, spp=256, bounces=8, sample_res=8),
}


def frame_seed(f):
    """u_seed of frame f; stays inside the reference's [0,10000) (dispatch.java:698)."""
    return (1234 + 7919 * f) % 10000


def make_params(W, H, sample_res=8, max_bounces=8, blur=0.001, focal_distance=1.0, auto_focus=1.0, screen_size=1.5, focal_length=1.0):
    """Parameters block, dispatch.java:191-205 <-> frag.glsl:39-52."""
    return np.array([screen_size, focal_length, W, H / float(W), sample_res, max_bounces, 0.0, blur, focal_distance, 1.0, 0.0, auto_focus], dtype=np.float32)


# ---------------------------------------------------------------------------------- OBJ writer
class Obj:
    def __init__(self):
        self.lines = []
        self.nv = 0
        self.nn = 0

    def group(self, name):
        self.lines.append(f"o {name}")

    def usemtl(self, name):
        self.lines.append(f"usemtl {name}")

    def v(self, p):
        self.lines.append("v %.9g %.9g %.9g" % tuple(p))
        self.nv += 1
        return self.nv

    def vn(self, n):
        self.lines.append("vn %.9g %.9g %.9g" % tuple(n))
        self.nn += 1
        return self.nn

    def vt(self, uv):
        self.lines.append("vt %.9g %.9g" % tuple(uv))
        self.nt = getattr(self, "nt", 0) + 1
        return self.nt

    def f(self, a, b, c, na, nb=None, nc=None):
        nb = na if nb is None else nb
        nc = na if nc is None else nc
        self.lines.append(f"f {a}//{na} {b}//{nb} {c}//{nc}")

    def quad(self, p0, p1, p2, p3, n):
        i = [self.v(p) for p in (p0, p1, p2, p3)]
        k = self.vn(n)
        self.f(i[0], i[1], i[2], k)
        self.f(i[0], i[2], i[3], k)

    def quad_uv(self, p0, p1, p2, p3, n, uv0=(0.05, 0.05), uv1=(0.95, 0.95)):
        """quad with texture coordinates (first vt.y must not be 0: the packer reads that as "no uv")"""
        i = [self.v(p) for p in (p0, p1, p2, p3)]
        k = self.vn(n)
        t = []
        for (a, b) in ((uv0[0], uv0[1]), (uv1[0], uv0[1]), (uv1[0], uv1[1]), (uv0[0], uv1[1])):
            self.lines.append("vt %.9g %.9g" % (a, b))
            self.nt = getattr(self, "nt", 0) + 1
            t.append(self.nt)
        self.lines.append(f"f {i[0]}/{t[0]}/{k} {i[1]}/{t[1]}/{k} {i[2]}/{t[2]}/{k}")
        self.lines.append(f"f {i[0]}/{t[0]}/{k} {i[2]}/{t[2]}/{k} {i[3]}/{t[3]}/{k}")

    def box(self, center, half, yrot):
        """axis-aligned box of half extents `half` rotated by yrot about +y, 12 triangles, flat normals."""
        c, s = math.cos(yrot), math.sin(yrot)

        def R(p):
            return (c * p[0] + s * p[2], p[1], -s * p[0] + c * p[2])

        def P(x, y, z):
            q = R((x * half[0], y * half[1], z * half[2]))
            return (q[0] + center[0], q[1] + center[1], q[2] + center[2])

        faces = [((1, 0, 0), [(1, -1, -1), (1, 1, -1), (1, 1, 1), (1, -1, 1)]), ((-1, 0, 0), [(-1, -1, 1), (-1, 1, 1), (-1, 1, -1), (-1, -1, -1)]),
                 ((0, 1, 0), [(-1, 1, -1), (-1, 1, 1), (1, 1, 1), (1, 1, -1)]), ((0, -1, 0), [(-1, -1, 1), (-1, -1, -1), (1, -1, -1), (1, -1, 1)]),
                 ((0, 0, 1), [(1, -1, 1), (1, 1, 1), (-1, 1, 1), (-1, -1, 1)]), ((0, 0, -1), [(-1, -1, -1), (-1, 1, -1), (1, 1, -1), (1, -1, -1)])]
        for n, corners in faces:
            self.quad(*[P(*q) for q in corners], R(n))

    def mesh(self, verts, normals, faces):
        base_v, base_n = self.nv, self.nn
        for p in verts:
            self.v(p)
        for n in normals:
            self.vn(n)
        for a, b, c in faces:
            self.lines.append(f"f {a+1+base_v}//{a+1+base_n} {b+1+base_v}//{b+1+base_n} {c+1+base_v}//{c+1+base_n}")

    def text(self):
        return "\n".join(self.lines) + "\n"


def icosphere(subdiv, center, radius):
    t = (1.0 + math.sqrt(5.0)) / 2.0
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t), (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    v = [np.array(p, dtype=np.float64) / np.linalg.norm(p) for p in v]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6), (7, 1, 8),
         (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    for _ in range(subdiv):
        cache, nf = {}, []

        def mid(a, b):
            key = (min(a, b), max(a, b))
            if key not in cache:
                m = v[a] + v[b]
                v.append(m / np.linalg.norm(m))
                cache[key] = len(v) - 1
            return cache[key]

        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        f = nf
    # tilt the unit sphere slightly so that no vertex normal has an exactly-zero component
    # (such normals take the flat-normal branch of frag.glsl:501)
    ax, ay = 0.1234, 0.2345
    Rx = np.array([[1, 0, 0], [0, math.cos(ax), -math.sin(ax)], [0, math.sin(ax), math.cos(ax)]])
    Ry = np.array([[math.cos(ay), 0, math.sin(ay)], [0, 1, 0], [-math.sin(ay), 0, math.cos(ay)]])
    n = [Ry @ (Rx @ p) for p in v]
    verts = [tuple(np.array(center) + radius * q) for q in n]
    return verts, [tuple(q) for q in n], f


def displaced_torus(nu, nv_, center, R, r, amp, seed):
    """~2*nu*nv triangles; radial displacement = seeded sum of sines; normals from the analytic torus."""
    rs = np.random.RandomState(seed)
    ku, kv, ph = rs.randint(1, 9, size=6), rs.randint(1, 9, size=6), rs.uniform(0, 2 * math.pi, size=6)
    verts, normals, faces = [], [], []
    for i in range(nu):
        u = 2 * math.pi * i / nu
        for j in range(nv_):
            w = 2 * math.pi * j / nv_
            d = sum(math.sin(ku[k] * u + kv[k] * w + ph[k]) for k in range(6)) / 6.0
            rr = r * (1.0 + amp * d)
            n = (math.cos(w) * math.cos(u), math.sin(w), math.cos(w) * math.sin(u))
            p = ((R + rr * math.cos(w)) * math.cos(u), rr * math.sin(w), (R + rr * math.cos(w)) * math.sin(u))
            verts.append((p[0] + center[0], p[1] + center[1], p[2] + center[2]))
            normals.append(n)
    for i in range(nu):
        for j in range(nv_):
            a, b = i * nv_ + j, ((i + 1) % nu) * nv_ + j
            c, d = ((i + 1) % nu) * nv_ + (j + 1) % nv_, i * nv_ + (j + 1) % nv_
            faces += [(a, b, c), (a, c, d)]
    return verts, normals, faces


# ---------------------------------------------------------------------------------- scenes
class Workload:
    """Everything the render call consumes: SSBO contents by binding point + texture 0 + config."""

    def __init__(self, name, W, H, buffers, sky, sample_res, max_bounces, info, textures=None):
        self.name, self.W, self.H = name, W, H
        self.buffers = buffers            # {binding: np.ndarray}: 0,1,2,3,4,5,7,10,11,12,13,14
        self.sky = sky                    # (h, w, 4) uint8, texture index 0
        self.textures = dict(textures or {})   # {index >= 1: (h, w, 4) uint8}: the rest of the bindless table (binding 15)
        self.sample_res, self.max_bounces = sample_res, max_bounces
        self.info = info

    def with_params(self, **kw):
        p = self.buffers[4].copy()
        names = ["screenSize", "focalLength", "resolution", "screenHratio", "SAMPLE_RES", "MAX_BOUNCES", "GAMMA", "BLUR", "FOCAL_DISTANCE", "RAYTRACING", "DEBUG", "AUTO_FOCUS"]
        for k, v in kw.items():
            p[names.index(k)] = v
        b = dict(self.buffers)
        b[4] = p
        return Workload(self.name, self.W, self.H, b, self.sky, int(p[4]), int(p[5]), self.info, self.textures)


GPU_BVH = None          # set to a device index to build every workload's BVHs with pt_build_bvh (libpt_hip.so) instead of the CPU builder


def _new_scene():
    sc = hostlib.Scene()
    if GPU_BVH is not None:
        sc.use_gpu_bvh_builder(GPU_BVH)
    return sc


def _finish(name, sc, W, H, cam, rot, sky_rgb, sample_res, max_bounces, **pk):
    bufs = sc.pack()
    bufs[0] = np.array(cam, dtype=np.float32)
    bufs[1] = np.array(rot, dtype=np.float32)
    bufs[2] = np.array([-1.0e6, -1.0e6, 0.0], dtype=np.float32)      # MOUSE_POS off-screen (frag.glsl:888)
    bufs[4] = make_params(W, H, sample_res, max_bounces, **pk)
    sky = np.array(sky_rgb, dtype=np.uint8)
    if sky.ndim == 1:
        sky = np.array([[list(sky_rgb) + [255]]], dtype=np.uint8)
    info = dict(triangles=sc.count("triangles"), nodes=sc.count("nodes"), objects=sc.count("objects"), max_depth=sc.count("max_depth"),
                max_leaf=sc.count("max_leaf"), materials=sc.count("materials"), ellipsoids=sc.count("ellipsoids"))
    return Workload(name, W, H, bufs, sky, sample_res, max_bounces, info)


def _cornell_materials(sc):
    """white/red/green diffuse, light Ke=15. Names carry no directory suffix (parentDirectory="")."""
    idx = {}
    for name, kd in (("white", (0.73, 0.73, 0.73)), ("red", (0.65, 0.05, 0.05)), ("green", (0.12, 0.45, 0.15))):
        idx[name] = sc.addMaterial(name)
        sc.setLastMtl("Kd", kd)
        sc.setLastMtl("Pr", 1)
    idx["light"] = sc.addMaterial("light")
    sc.setLastMtl("Kd", (0.78, 0.78, 0.78))
    sc.setLastMtl("Ke", (15, 15, 15))
    return idx


def _cornell_room(o, boxes=True):
    # room: x in [-1,1], y in [0,2], z in [-1,1]; open towards -z (camera side)
    o.group("walls")
    o.usemtl("white")
    o.quad((-1, 0, -1), (1, 0, -1), (1, 0, 1), (-1, 0, 1), (0, 1, 0))          # floor
    o.quad((-1, 2, -1), (-1, 2, 1), (1, 2, 1), (1, 2, -1), (0, -1, 0))         # ceiling
    o.quad((-1, 0, 1), (1, 0, 1), (1, 2, 1), (-1, 2, 1), (0, 0, -1))           # back
    o.usemtl("red")
    o.quad((1, 0, -1), (1, 2, -1), (1, 2, 1), (1, 0, 1), (-1, 0, 0))           # +x wall (image left: x is mirrored, frag.glsl:894)
    o.usemtl("green")
    o.quad((-1, 0, -1), (-1, 0, 1), (-1, 2, 1), (-1, 2, -1), (1, 0, 0))        # -x wall
    o.group("light")
    o.usemtl("light")
    o.quad((-0.25, 1.98, -0.25), (0.25, 1.98, -0.25), (0.25, 1.98, 0.25), (-0.25, 1.98, 0.25), (0, -1, 0))
    if boxes:
        o.group("tallbox")
        o.usemtl("white")
        o.box((0.35, 0.6, 0.3), (0.3, 0.6, 0.3), 0.3)
        o.group("shortbox")
        o.usemtl("white")
        o.box((-0.35, 0.3, -0.3), (0.3, 0.3, 0.3), -0.3)


CORNELL_CAM = (0.0, 1.0, -1.6)
CORNELL_ROT = (0.0, 0.0, 0.0)


def c1_spheres(W=256, H=256, sample_res=4, max_bounces=4):
    """C1 'built-in sphere scene': 3 addEllipsoid over a ground quad (pattern of dispatch.java:245,264), 1x1 sky."""
    sc = _new_scene()
    sc.addMaterial("default")
    sc.setLastMtl("Kd", (0.8, 0.8, 0.8))
    sc.setLastMtl("Pr", 1)
    sc.addMaterial("metal")
    sc.setLastMtl("Pm", 1)
    sc.setLastMtl("Pr", 0.2)
    o = Obj()
    o.group("ground")
    o.quad((-6, 0, -2), (6, 0, -2), (6, 0, 10), (-6, 0, 10), (0, 1, 0))
    sc.addObjectText(o.text(), 0, parentDirectory="")
    sc.addEllipsoid((0.0, 0.5, 3.0), 1, 0, 0.5, 0)
    sc.addEllipsoid((1.1, 0.4, 2.5), 1, 0, 0.4, 1)
    sc.addEllipsoid((-1.0, 0.3, 2.2), (1.0, 2.0, 1.0), 0, 0.3, 0)
    return _finish("C1", sc, W, H, (0.0, 0.8, 0.0), (0.1, 0.0, 0.0), (153, 179, 230), sample_res, max_bounces)


def c2_cornell(W=1280, H=720, sample_res=8, max_bounces=8):
    """C2 diffuse-only Cornell box: 5 walls + light quad + 2 boxes = 36 triangles, black 1x1 sky."""
    sc = _new_scene()
    _cornell_materials(sc)
    o = Obj()
    _cornell_room(o)
    sc.addObjectText(o.text(), 0, parentDirectory="")
    return _finish("C2", sc, W, H, CORNELL_CAM, CORNELL_ROT, (0, 0, 0), sample_res, max_bounces)


def c3_glass_metal(W=1920, H=1080, sample_res=8, max_bounces=8, subdiv=3):
    """C3: Cornell room + glass and metal icospheres (1280 triangles each at subdiv 3, smooth vn)."""
    sc = _new_scene()
    _cornell_materials(sc)
    sc.addMaterial("glass")
    sc.setLastMtl("Tr", 0.9)
    sc.setLastMtl("Ni", 1.5)
    sc.setLastMtl("Pr", 1)
    sc.setLastMtl("Tf", (0.2, 0.05, 0.05))
    sc.setLastMtl("Density", 1)
    sc.addMaterial("metal")
    sc.setLastMtl("Pm", 1)
    sc.setLastMtl("Pr", 0.1)
    sc.setLastMtl("Kd", (0.9, 0.8, 0.5))
    o = Obj()
    _cornell_room(o, boxes=False)
    o.group("glass_sphere")
    o.usemtl("glass")
    o.mesh(*icosphere(subdiv, (-0.42, 0.45, -0.25), 0.45))
    o.group("metal_sphere")
    o.usemtl("metal")
    o.mesh(*icosphere(subdiv, (0.45, 0.5, 0.35), 0.5))
    sc.addObjectText(o.text(), 0, parentDirectory="")
    return _finish("C3", sc, W, H, CORNELL_CAM, CORNELL_ROT, (0, 0, 0), sample_res, max_bounces)


def c4_mesh(W=1920, H=1080, sample_res=8, max_bounces=8, nu=224, nv=224, seed=4):
    """C4: Cornell room + one seeded displaced-torus mesh of 2*nu*nv (= 100 352) triangles in ONE `o` group
    (one BVH, like the reference's single-object dragon, dispatch.java:257)."""
    sc = _new_scene()
    _cornell_materials(sc)
    sc.addMaterial("clay")
    sc.setLastMtl("Kd", (0.7, 0.55, 0.4))
    sc.setLastMtl("Pr", 1)
    o = Obj()
    _cornell_room(o, boxes=False)
    o.group("torus")
    o.usemtl("clay")
    v, n, f = displaced_torus(nu, nv, (0.0, 0.0, 0.0), 0.55, 0.22, 0.25, seed)
    # stand the torus up, tilted, in the middle of the room
    a, b = 1.0, 0.4
    Rx = np.array([[1, 0, 0], [0, math.cos(a), -math.sin(a)], [0, math.sin(a), math.cos(a)]])
    Ry = np.array([[math.cos(b), 0, math.sin(b)], [0, 1, 0], [-math.sin(b), 0, math.cos(b)]])
    M = Ry @ Rx
    v = [tuple(M @ np.array(p) + np.array((0.0, 0.85, 0.1))) for p in v]
    n = [tuple(M @ np.array(q)) for q in n]
    o.mesh(v, n, f)
    sc.addObjectText(o.text(), 0, parentDirectory="")
    return _finish("C4", sc, W, H, CORNELL_CAM, CORNELL_ROT, (0, 0, 0), sample_res, max_bounces)


def c5_clearcoat_sss(W=3840, H=2160, sample_res=8, max_bounces=16, subdiv=4):
    """C5: Cornell room + clearcoat and subsurface meshes, plus the reference's "test" material (dispatch.java:228-239)."""
    sc = _new_scene()
    _cornell_materials(sc)
    sc.addMaterial("coat")
    sc.setLastMtl("Kd", (0.1, 0.2, 0.7))
    sc.setLastMtl("Ks", (0.9, 0.9, 0.9))
    sc.setLastMtl("Pc", 0.5)
    sc.setLastMtl("Pcr", 0.1)
    sc.setLastMtl("Pr", 1)
    sc.addMaterial("sss")
    sc.setLastMtl("Kd", (0.8, 0.45, 0.5))
    sc.setLastMtl("subsurface", 0.5)
    sc.setLastMtl("subsurfaceColor", (0.45, 0.8, 0.5))
    sc.setLastMtl("subsurfaceRadius", (1, 1, 1))
    sc.setLastMtl("Pr", 1)
    sc.addMaterial("test")                      # dispatch.java:228-239
    sc.setLastMtl("Kd", (0.8, 0.45, 0.5))
    sc.setLastMtl("Ks", (0.5, 0.5, 0.5))
    sc.setLastMtl("Ni", 1.45)
    sc.setLastMtl("Pr", 1)
    sc.setLastMtl("Pc", 0.0)
    sc.setLastMtl("Pcr", 0.0)
    sc.setLastMtl("Tr", 0.7)
    sc.setLastMtl("subsurface", 0)
    sc.setLastMtl("subsurfaceColor", (0.45, 0.8, 0.5))
    sc.setLastMtl("subsurfaceRadius", (1, 1, 1))
    sc.setLastMtl("Density", 0.1)
    o = Obj()
    _cornell_room(o, boxes=False)
    o.group("coat_sphere")
    o.usemtl("coat")
    o.mesh(*icosphere(subdiv, (0.45, 0.5, 0.3), 0.5))
    o.group("sss_sphere")
    o.usemtl("sss")
    o.mesh(*icosphere(subdiv, (-0.45, 0.4, -0.2), 0.4))
    o.group("test_sphere")
    o.usemtl("test")
    o.mesh(*icosphere(subdiv - 1, (0.0, 1.3, 0.0), 0.3))
    sc.addObjectText(o.text(), 0, parentDirectory="")
    return _finish("C5", sc, W, H, CORNELL_CAM, CORNELL_ROT, (0, 0, 0), sample_res, max_bounces)


def t1_textured(W=96, H=54, sample_res=8, max_bounces=8):
    """Texture-map workload (not a BASELINE config): Cornell room whose floor has a checker map_Kd, whose back
    wall carries map_Ke + map_Ks, a box with map_Pr / map_Pm / map_Pc / map_Tr, and a panel shaded through a raw-texel map_bump."""
    rs = np.random.RandomState(11)

    def tex(h, w, fn):
        a = np.zeros((h, w, 4), np.uint8)
        for j in range(h):
            for i in range(w):
                a[j, i, :3] = fn(i, j)
        a[..., 3] = 255
        return a

    textures = {
        1: tex(8, 8, lambda i, j: (230, 230, 230) if (i + j) % 2 else (40, 60, 200)),                      # checker albedo
        2: tex(4, 16, lambda i, j: (255, 200, 120) if i % 4 == 0 else (0, 0, 0)),                           # emissive stripes
        3: tex(5, 7, lambda i, j: tuple(int(v) for v in rs.randint(0, 256, 3))),                              # noise (Ks / Pr / Pm / Pc / Tr source)
        4: tex(2, 2, lambda i, j: (60, 230, 90)),                                                             # "normal" texels, used raw (frag.glsl:827)
    }
    sc = _new_scene()
    _cornell_materials(sc)
    sc.addMaterial("floor"); sc.setLastMtl("Kd", (0.9, 0.9, 0.9)); sc.setLastMtl("Pr", 1); sc.setLastMtl("map_Kd", 1)
    sc.addMaterial("glow"); sc.setLastMtl("Kd", (0.5, 0.5, 0.5)); sc.setLastMtl("Pr", 1); sc.setLastMtl("map_Ke", 2); sc.setLastMtl("map_Ks", 3); sc.setLastMtl("Pc", 0.3)
    sc.addMaterial("mixed"); sc.setLastMtl("Kd", (0.8, 0.7, 0.6)); sc.setLastMtl("map_Pr", 3); sc.setLastMtl("map_Pm", 3); sc.setLastMtl("map_Pc", 3)
    sc.setLastMtl("map_Tr", 3); sc.setLastMtl("Ni", 1.3); sc.setLastMtl("map_Ka", 1); sc.setLastMtl("Ka", (0.1, 0.1, 0.1))
    sc.addMaterial("bumped"); sc.setLastMtl("Kd", (0.7, 0.7, 0.7)); sc.setLastMtl("Pr", 1); sc.setLastMtl("map_bump", 4)
    o = Obj()
    o.group("room")
    o.usemtl("floor")
    o.quad_uv((-1, 0, -1), (1, 0, -1), (1, 0, 1), (-1, 0, 1), (0, 1, 0), (0.05, 0.05), (3.95, 3.95))
    o.usemtl("white")
    o.quad((-1, 2, -1), (-1, 2, 1), (1, 2, 1), (1, 2, -1), (0, -1, 0))
    o.usemtl("glow")
    o.quad_uv((-1, 0, 1), (1, 0, 1), (1, 2, 1), (-1, 2, 1), (0, 0, -1))
    o.usemtl("red")
    o.quad((1, 0, -1), (1, 2, -1), (1, 2, 1), (1, 0, 1), (-1, 0, 0))
    o.usemtl("green")
    o.quad((-1, 0, -1), (-1, 0, 1), (-1, 2, 1), (-1, 2, -1), (1, 0, 0))
    o.group("light")
    o.usemtl("light")
    o.quad((-0.25, 1.98, -0.25), (0.25, 1.98, -0.25), (0.25, 1.98, 0.25), (-0.25, 1.98, 0.25), (0, -1, 0))
    o.group("panel")
    o.usemtl("bumped")
    o.quad_uv((-0.9, 0.2, 0.2), (-0.3, 0.2, -0.4), (-0.3, 1.2, -0.4), (-0.9, 1.2, 0.2), (0.7, 0.1, -0.7))
    o.usemtl("mixed")
    o.quad_uv((0.2, 0.1, 0.3), (0.9, 0.1, -0.2), (0.9, 1.0, -0.2), (0.2, 1.0, 0.3), (-0.58, 0.05, -0.81))
    sc.addObjectText(o.text(), 0, parentDirectory="")
    wl = _finish("T1", sc, W, H, CORNELL_CAM, CORNELL_ROT, (30, 40, 60), sample_res, max_bounces)
    wl.textures = textures
    return wl


def asset_workload(directory, W, H, cam=CORNELL_CAM, rot=CORNELL_ROT, sky=(30, 40, 60), sample_res=8, max_bounces=8, name="asset",
                   scale=1.0, shift=0.0, rotate=0.0, **pk):
    """The reference's way of loading a model (dispatch.java:219-229 + :869-882): texture 0 is the sky, then
    scene.addObject(<directory>) parses every .mtl (registering the map files) and every .obj in it; the decoded images fill
    the rest of the texture table.  `sky` is an RGB triple or a path to an image."""
    sc = _new_scene()
    sky_img = sky
    if isinstance(sky, str):
        from PIL import Image
        sc.addTexture(sky, "skybox.png")
        sky_img = np.asarray(Image.open(sky).convert("RGBA"), dtype=np.uint8).copy()
    else:
        sc.addTexture("", "skybox.png")
    sc.addMaterial("default")
    sc.addObject(directory, 0, scale, shift, rotate)
    wl = _finish(name, sc, W, H, cam, rot, sky_img, sample_res, max_bounces, **pk)
    wl.textures = {i: a for i, a in sc.load_textures().items() if i >= 1}
    return wl


def c6_many_objects(W=1920, H=1080, sample_res=8, max_bounces=8, groups=64, nu=28, nv=28, seed=6):
    """C6: the shape of the scenes the reference's author lists (dispatch.java:245-264: multi-group OBJs with dozens of `o` groups — one BVH
    each, frag.glsl:563-577 loops over all of them — next to addEllipsoid calls): the Cornell room's two groups + (groups - 2) seeded
    displaced tori of 2*nu*nv triangles each in a 4x4x4 lattice (64 groups: 97 228 triangles), materials cycling through diffuse / metal /
    glass / clearcoat, plus 3 ellipsoids (one stretched, one rotated)."""
    sc = _new_scene()
    _cornell_materials(sc)
    names = []
    for k, (kd, extra) in enumerate([((0.7, 0.55, 0.4), {}), ((0.9, 0.8, 0.5), dict(Pm=1, Pr=0.1)), ((0.9, 0.9, 0.9), dict(Tr=0.9, Ni=1.5, Tf=(0.05, 0.2, 0.05), Density=1)),
                                     ((0.1, 0.2, 0.7), dict(Ks=(0.9, 0.9, 0.9), Pc=0.5, Pcr=0.1)), ((0.3, 0.6, 0.35), {}), ((0.6, 0.3, 0.6), dict(Pm=0.5, Pr=0.4))]):
        names.append(f"m{k}")
        sc.addMaterial(names[-1])
        sc.setLastMtl("Kd", kd)
        sc.setLastMtl("Pr", 1)
        for key, val in extra.items():
            sc.setLastMtl(key, val)
    o = Obj()
    _cornell_room(o, boxes=False)
    rs = np.random.RandomState(seed)
    cells = [(i, j, k) for k in range(4) for j in range(4) for i in range(4)]
    for g in range(groups - 2):
        i, j, k = cells[g % len(cells)]
        centre = np.array((-0.75 + 0.5 * i, 0.25 + 0.48 * j, -0.75 + 0.5 * k)) + rs.uniform(-0.04, 0.04, size=3)
        v, n, f = displaced_torus(nu, nv, (0.0, 0.0, 0.0), 0.13, 0.055, 0.25, seed * 1000 + g)
        a, b = rs.uniform(0, math.pi, size=2)
        Rx = np.array([[1, 0, 0], [0, math.cos(a), -math.sin(a)], [0, math.sin(a), math.cos(a)]])
        Ry = np.array([[math.cos(b), 0, math.sin(b)], [0, 1, 0], [-math.sin(b), 0, math.cos(b)]])
        M = Ry @ Rx
        o.group(f"torus{g}")
        o.usemtl(names[g % len(names)])
        o.mesh([tuple(M @ np.array(p) + centre) for p in v], [tuple(M @ np.array(q)) for q in n], f)
    sc.addObjectText(o.text(), 0, parentDirectory="")
    sc.addEllipsoid((0.0, 1.0, 0.0), 1, 0, 0.16, 5)
    sc.addEllipsoid((0.5, 0.75, -0.5), (1.0, 2.5, 1.0), 0, 0.12, 4)
    sc.addEllipsoid((-0.5, 1.25, 0.5), (2.0, 1.0, 1.0), (0.3, 0.2, 0.1), 0.14, 7)
    return _finish("C6", sc, W, H, CORNELL_CAM, CORNELL_ROT, (0, 0, 0), sample_res, max_bounces)


def equirect_sky(h, w, seed=9):
    """A synthetic equirect sky image (h, w, 4) uint8 in place of the reference's thatch_chapel_4k.png (dispatch.java:221, not in the repository): smooth
    gradients + seeded noise + a small "sun", every texel different from its neighbours, so that the bilinear REPEAT sampler (frag.glsl:235-242) is
    exercised on every miss."""
    rs = np.random.RandomState(seed)
    y, x = np.mgrid[0:h, 0:w]
    img = np.zeros((h, w, 4), np.float64)
    img[..., 0] = 90 + 80 * np.sin(2 * np.pi * x / w) + 40 * y / h
    img[..., 1] = 110 + 60 * np.cos(4 * np.pi * x / w) * (1 - y / h)
    img[..., 2] = 200 - 120 * y / h
    img[..., :3] += rs.uniform(-12, 12, size=(h, w, 3))
    sun = (x - 0.3 * w) ** 2 + (y - 0.25 * h) ** 2 < (0.02 * w) ** 2
    img[sun, :3] = 255
    img[..., 3] = 255
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


BUILDERS = {"C1": c1_spheres, "C2": c2_cornell, "C3": c3_glass_metal, "C4": c4_mesh, "C5": c5_clearcoat_sss, "C6": c6_many_objects}


def build(name, W=None, H=None, **kw):
    if name == "T1":
        return t1_textured(W or 96, H or 54, **kw)
    cfg = CONFIGS[name]
    W = cfg["W"] if W is None else W
    H = cfg["H"] if H is None else H
    kw.setdefault("sample_res", cfg["sample_res"])
    kw.setdefault("max_bounces", cfg["bounces"])
    return BUILDERS[name](W, H, **kw)
