"""CPU: pins the oracle (the reference ships no tests or golden vectors).

K1  integer-exact RNG vectors (frag.glsl:686-694), SURVEY.md §8(c) table
K2-K9 analytic known-answer scenes whose results follow from the cited shader lines
plus accuracy of the numeric contract's elementary functions against float64 libm.
"""
import numpy as np
import pytest


# ------------------------------------------------------------------------------------------ K1
RNG_KAT = {  # start state -> [(state, result, random() bits)] x 4     
    0: [(2891336453, 129708002, 0x3cf765fc), (1192405134, 582399676, 0x3e0adadb), (568162667, 1006035121, 0x3e6fdb83), (878960812, 1462727737, 0x3eae5ee8)],
    1: [(3639132858, 2831084092, 0x3f28beea), (1098935943, 645514520, 0x3e19e714), (3968020856, 2544563910, 0x3f17aaf7), (3079081181, 710022002, 0x3e29484e)],
    12345: [(258269778, 4099845390, 0x3f745ead), (2661444863, 430018922, 0x3dcd0c8b), (1397089360, 1600717359, 0x3ebed204), (4286703509, 2315322805, 0x3f0a0106)],
    2083598: [(3224395243, 501798100, 0x3def46a7), (4265013036, 2921031906, 0x3f2e1b69), (3916618017, 2093285126, 0x3ef989fe), (2655659610, 56501086, 0x3c5788d8)],
}


@pytest.mark.parametrize("start", sorted(RNG_KAT))
def test_k1_rng_vectors(oracle, start):
    st = start
    for exp_state, exp_res, exp_bits in RNG_KAT[start]:
        st, res, rnd = oracle.rng(st, 1)
        assert st == exp_state and int(res[0]) == exp_res and int(rnd.view(np.uint32)[0]) == exp_bits


def test_k1_random_reaches_exactly_one(oracle):
    # random() = float(r)/4294967295.0 with the literal rounding to 2^32: results >= 4294967168 give exactly 1.0f
    assert np.float32(4294967168) / np.float32(4294967295.0) == np.float32(1.0)
    assert np.float32(4294967167) / np.float32(4294967295.0) < np.float32(1.0)
    assert np.float32(69.420).view(np.uint32) == 0x428AD70A


# ------------------------------------------------------------------------------------------ numeric contract
def _ulp_err(got, ref64):
    ref32 = ref64.astype(np.float32)
    ulp = np.abs(np.spacing(ref32)).astype(np.float64)
    return np.abs(got.astype(np.float64) - ref64) / np.maximum(ulp, 1e-300)


@pytest.mark.parametrize("fn,npfn,lo,hi,tol", [
    ("sin", np.sin, -10.0, 10.0, 2.5), ("cos", np.cos, -10.0, 10.0, 2.5), ("log", np.log, 1e-9, 50.0, 1.5), ("exp", np.exp, -80.0, 80.0, 2.0),
    ("asin", np.arcsin, -1.0, 1.0, 3.0)])
def test_math_accuracy(oracle, fn, npfn, lo, hi, tol):
    x = np.random.RandomState(0).uniform(lo, hi, 200000).astype(np.float32)
    got = oracle.math(fn, x)
    ref = npfn(x.astype(np.float64))
    if fn in ("sin", "cos"):   # absolute error near zeros of sin/cos: compare against 1 ulp of 1.0 scaled
        assert np.max(np.abs(got - ref)) < 2.5e-7
    else:
        assert np.max(_ulp_err(got, ref)) <= tol


def test_math_atan2_accuracy_and_specials(oracle):
    rs = np.random.RandomState(1)
    y, x = rs.normal(size=100000).astype(np.float32), rs.normal(size=100000).astype(np.float32)
    got = oracle.math("atan2", y, x)
    assert np.max(np.abs(got - np.arctan2(y.astype(np.float64), x.astype(np.float64)))) < 6e-7
    assert oracle.math("log", np.array([0.0], np.float32))[0] == -np.inf
    assert np.isnan(oracle.math("log", np.array([-1.0], np.float32))[0])
    assert np.isnan(oracle.math("asin", np.array([1.5], np.float32))[0])
    assert oracle.math("exp", np.array([-200.0], np.float32))[0] == 0.0 and oracle.math("exp", np.array([100.0], np.float32))[0] == np.inf
    assert oracle.math("atan2", np.array([0.0], np.float32), np.array([0.0], np.float32))[0] == 0.0


def test_rotate_matches_java_convention(oracle):
    """rotate() == Rz*Ry*Rx (X first), the same convention as Java vec.rotate (frag.glsl:244-283, dispatch.java:1157-1191)"""
    rs = np.random.RandomState(2)
    for _ in range(50):
        p = rs.normal(size=3); a = rs.uniform(-3, 3, size=3)
        if _ % 5 == 0:
            a[2] = 0.0
        cx, sx, cy, sy, cz, sz = np.cos(a[0]), np.sin(a[0]), np.cos(a[1]), np.sin(a[1]), np.cos(a[2]), np.sin(a[2])
        Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]]); Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]]); Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
        assert np.allclose(oracle.rotate(p, a), Rz @ Ry @ Rx @ p, atol=2e-6)


# ------------------------------------------------------------------------------------------ scenes for K2-K9
def _scene(pt, W, H, tris_obj=None, mats=(), ellipsoids=(), sky=(10, 20, 30), cam=(0, 0, 0), rot=(0, 0, 0), **params):
    sc = pt.hostlib.Scene()
    for m in mats:
        sc.addMaterial(m.get("name", "m"))
        for k, v in m.items():
            if k != "name":
                sc.setLastMtl(k, v)
    if tris_obj:
        sc.addObjectText(tris_obj, 0, parentDirectory="")
    for e in ellipsoids:
        sc.addEllipsoid(*e)
    b = sc.pack()
    b[0] = np.array(cam, np.float32); b[1] = np.array(rot, np.float32); b[2] = np.array([-1e6, -1e6, 0], np.float32)
    kw = dict(sample_res=2, max_bounces=4, blur=0.0)
    kw.update(params)
    b[4] = pt.scenes.make_params(W, H, **kw)
    skyarr = np.array(sky, np.uint8)
    if skyarr.ndim == 1:
        skyarr = np.array([[list(sky) + [255]]], np.uint8)
    return b, skyarr


BIG_TRI = "o tri\nv -50 -40 {z}\nv 50 -40 {z}\nv 0 60 {z}\nv -50 -40 {z2}\nv 50 -40 {z2}\nv 0 60 {z2}\nvn {n}\nf 1//1 2//1 3//1\nf 4//1 5//1 6//1\n"


def _render(oracle, b, sky, W, H, frames=1, first=1):
    sc = oracle.Scene(b, sky)
    frame = None
    for f in range(frames):
        frame, cnt = oracle.render(sc, W, H, first + f, 1000 + 37 * f, frame)
    return frame


def test_k2_empty_scene_is_sky(pt, oracle):
    W, H = 16, 12
    b, sky = _scene(pt, W, H, mats=[{}], sky=(51, 102, 204))
    fr = _render(oracle, b, sky, W, H)
    c = np.array([51, 102, 204], np.float32) / np.float32(255)
    # the four bilinear weights sum to 1 up to rounding -> within 1 ulp of the texel
    assert np.allclose(fr[..., :3], c, rtol=3e-7, atol=0) and np.all(fr[..., 3] == 1.0)


def test_k3_sky_mapping(pt, oracle):
    """uv = (0.5 + atan(z,x)/(2*3.14159), 0.5 - asin(y)/3.14159) of the normalised primary direction (frag.glsl:235-242)"""
    W, H = 24, 16
    sky = np.zeros((2, 4, 4), np.uint8)
    for j in range(2):
        for i in range(4):
            sky[j, i] = (40 * i + 10, 100 * j + 20, 7 * (i + 4 * j), 255)
    b, _ = _scene(pt, W, H, mats=[{}], blur=0.0, auto_focus=0.0)
    fr = _render(oracle, b, sky, W, H)
    ratio = H / W
    for (x, y) in [(0, 0), (5, 3), (23, 15), (12, 8), (2, 14)]:
        d = np.array([-(2 * (x + .5) / W - 1) * 1.5, (2 * (y + .5) / H - 1) * ratio * 1.5, 1.0])
        d /= np.linalg.norm(d)
        u = 0.5 + np.arctan2(d[2], d[0]) / (2 * 3.14159); v = 0.5 - np.arcsin(d[1]) / 3.14159
        fu, fv = u * 4 - 0.5, v * 2 - 0.5
        i0, j0 = int(np.floor(fu)), int(np.floor(fv)); a, bb = fu - i0, fv - j0
        tex = lambda i, j: sky[j % 2, i % 4, :3].astype(np.float64) / 255
        exp = (1 - a) * (1 - bb) * tex(i0, j0) + a * (1 - bb) * tex(i0 + 1, j0) + (1 - a) * bb * tex(i0, j0 + 1) + a * bb * tex(i0 + 1, j0 + 1)
        assert np.allclose(fr[y, x, :3], exp, atol=2e-5), (x, y)


@pytest.mark.parametrize("normal,branch", [("0 0 -1", "flat n2 (Q-4)"), ("0.3 0.2 -0.9", "smooth (Q-3)")])
def test_k4_convex_diffuse_is_kd_times_sky(pt, oracle, normal, branch):
    W, H = 16, 12
    kd = (0.5, 0.25, 0.75)
    b, sky = _scene(pt, W, H, BIG_TRI.format(z=5, z2=500, n=normal), mats=[dict(Kd=kd, Pr=1)], sky=(255, 255, 255))
    fr = _render(oracle, b, sky, W, H)
    # every pixel hits the big triangle; the bounce ray leaves a convex scene and picks up the constant sky
    # -> Kd * sky, independent of the RNG.  (A near-tangent bounce can re-hit the plane it starts on: the
    # reference offsets the origin along d, not N (frag.glsl:549) -> a few pixels get one more Kd factor.)
    img = fr[..., :3].reshape(-1, 3)
    good = np.all(np.abs(img / np.array(kd, np.float32) - 1.0) < 1e-6, axis=1)
    assert good.mean() > 0.95
    assert np.all(img <= np.array(kd, np.float32) * (1 + 1e-6))


def test_k5_emission_added_before_throughput(pt, oracle):
    W, H = 8, 6
    b, sky = _scene(pt, W, H, BIG_TRI.format(z=5, z2=500, n="0 0 -1"), mats=[dict(Kd=(0.5, 0.5, 0.5), Ke=(3, 2, 1), Pr=1)], sky=(0, 0, 0))
    fr = _render(oracle, b, sky, W, H)
    assert np.array_equal(fr[..., :3], np.broadcast_to(np.array([3, 2, 1], np.float32), fr[..., :3].shape))   # incLight += Ke*col with col = 1 (:865)


def test_k6_mirror_multiplies_by_kd_not_ks(pt, oracle):
    """Pr=0, Pm=1: reflectionWeight = 1 -> winType 1; throughput uses Kd (Q-9, frag.glsl:844,873)"""
    W, H = 8, 6
    kd, ks = (0.2, 0.4, 0.6), (0.9, 0.9, 0.9)
    b, sky = _scene(pt, W, H, BIG_TRI.format(z=5, z2=500, n="0 0 -1"), mats=[dict(Kd=kd, Ks=ks, Pr=0, Pm=1)], sky=(255, 255, 255))
    fr = _render(oracle, b, sky, W, H)
    assert np.allclose(fr[..., :3], np.array(kd, np.float32), rtol=1e-6)


def test_k7_accumulation_and_frame_counter(pt, oracle):
    W, H = 8, 6
    b, sky = _scene(pt, W, H, BIG_TRI.format(z=5, z2=500, n="0 0 -1"), mats=[dict(Kd=(0.5, 0.5, 0.5), Ke=(1, 1, 1), Pr=1)], sky=(0, 0, 0))
    sc = oracle.Scene(b, sky)
    fr = np.full((H, W, 4), 7.0, np.float32)          # garbage: frame count 1 must overwrite it
    oracle.render(sc, W, H, 1, 5, fr)
    assert np.all(fr[..., 3] == 1.0) and np.all(fr[..., 0] == 1.0)
    oracle.render(sc, W, H, 2, 6, fr); oracle.render(sc, W, H, 3, 7, fr)
    assert np.all(fr[..., 3] == 3.0) and np.all(fr[..., 0] == 3.0)
    oracle.render(sc, W, H, 0, 8, fr)                 # frame 0 adds (Q-13)
    assert np.all(fr[..., 3] == 4.0)


def test_k9_cutoff_before_multiply(pt, oracle):
    """two facing dark planes: the path returns as soon as length(col) < 0.1, tested BEFORE the current hit's multiply (:866)"""
    W, H = 8, 6
    obj = ("o a\nv -500 -400 5\nv 500 -400 5\nv 0 600 5\nv -500 -400 6\nv 500 -400 6\nv 0 600 6\nvn 0 0 -1\nf 1//1 2//1 3//1\nf 4//1 5//1 6//1\n"
           "o b\nv -500 -400 -5\nv 500 -400 -5\nv 0 600 -5\nv -500 -400 -6\nv 500 -400 -6\nv 0 600 -6\nvn 0 0 1\nf 7//2 8//2 9//2\nf 10//2 11//2 12//2\n")
    kd = 0.2     # |col| after k hits = sqrt(3)*0.2^k: 1.73, 0.346, 0.069 -> third hit returns
    b, sky = _scene(pt, W, H, obj, mats=[dict(Kd=(kd, kd, kd), Ke=(1, 1, 1), Pr=1)], sky=(0, 0, 0), max_bounces=50, sample_res=1)
    sc = oracle.Scene(b, sky)
    fr, cnt = oracle.render(sc, W, H, 1, 3)
    c = dict(zip(oracle.COUNTERS, cnt.tolist()))
    assert c["segments"] == 3 * W * H
    exp = np.float32(1.0) + np.float32(kd) + np.float32(kd) * np.float32(kd)
    escaped = np.float32(1.0) + np.float32(kd)          # a grazing third segment can slip out between the planes (black sky)
    v = fr[..., :3]
    assert np.all(np.isclose(v, exp, rtol=1e-6) | np.isclose(v, escaped, rtol=1e-6)) and np.isclose(v, exp, rtol=1e-6).mean() > 0.9


def test_q6_ellipsoid_from_inside_returns_negative_root(pt, oracle):
    W, H = 8, 8
    b, sky = _scene(pt, W, H, mats=[dict(Kd=(0.5, 0.5, 0.5), Pr=1)], ellipsoids=[((0, 0, 0), 1, 0, 2.0, 0)])
    sc = oracle.Scene(b, sky)
    code, out = oracle.ray_scene(sc, (0, 0, 0), (0, 0, 1))
    assert code == (3 << 24) and out[0] < 0 and abs(out[0] + 2.0) < 1e-3       # near root, behind the origin
    code, out = oracle.ray_scene(sc, (0, 0, -5), (0, 0, 1))
    assert code == (3 << 24) and abs(out[0] - 3.0) < 1e-3 and np.allclose(out[4:7], (0, 0, -1), atol=1e-4)


def test_debug_heatmap_known_answer(pt, oracle):
    """DEBUG != 0 (frag.glsl:916-918 -> debugRayScene :539-547): one quad = one BVH whose root is an inner node with two leaves.  Every
    ray pops the root (boxTests = 2 -> blue = exp(0.01*(2-200))); it then pops 0, 1 or 2 leaves (red = 0.1*(0.1 per leaf) + exp(0.02*(0-150)),
    triTests is never incremented in the shader); green = 0; no random numbers: the seed does not matter and frames just add up."""
    W, H = 24, 18
    wl = pt.scenes.build("C2", W, H)          # any workload's camera/params; the scene below replaces its geometry
    sc = pt.hostlib.Scene(); sc.addMaterial("m")
    sc.addObjectText("o quad\nvn 0 0 -1\nv -0.5 0.5 1\nv 0.5 0.5 1\nv 0.5 1.5 1\nv -0.5 1.5 1\nf 1//1 2//1 3//1\nf 1//1 3//1 4//1\n", 0)
    b = dict(wl.buffers); b.update(sc.pack())
    b[4] = wl.buffers[4].copy(); b[4][10] = 1.0
    s = oracle.Scene(b, wl.sky)
    one, _ = oracle.render(s, W, H, 1, 111, nthreads=2)
    other, _ = oracle.render(s, W, H, 1, 9999, nthreads=1)
    assert np.array_equal(one, other)
    e3, e198 = np.exp(np.float32(-3.0)), np.exp(np.float32(-1.98))
    assert np.allclose(one[..., 2], e198, rtol=2e-6) and np.all(one[..., 1] == 0) and np.all(one[..., 3] == 1)
    red = np.unique(np.round((one[..., 0] - e3) * 100).astype(int))
    assert set(red.tolist()) <= {0, 1, 2} and len(red) >= 2          # some rays miss both leaf boxes, some enter one or both
    two = one.copy()
    oracle.render(s, W, H, 2, 5, two, nthreads=2)
    assert np.array_equal(two[..., :3], one[..., :3] + one[..., :3]) and np.all(two[..., 3] == 2)


def test_oracle_threads_and_strides_agree(pt, oracle):
    wl = pt.scenes.build("C3", 48, 27)
    sc = oracle.Scene.from_workload(wl)
    a, ca = oracle.render(sc, 48, 27, 1, 77, nthreads=1)
    b, cb = oracle.render(sc, 48, 27, 1, 77, nthreads=5)
    assert np.array_equal(a, b) and np.array_equal(ca, cb)
    c, _ = oracle.render(sc, 48, 27, 1, 77, xs=4, ys=3, nthreads=2)
    assert np.array_equal(c[::3, ::4], a[::3, ::4]) and np.all(c[1::3] == 0)


def test_n4_display_path_and_java_signed_byte_quirk(oracle):
    """screenshot path (dispatch.java:804-833): total/frameCount -> UNORM8 -> (r<<16)+(g<<8)+b with signed Java bytes -> flip"""
    fr = np.zeros((2, 3, 4), np.float32)
    fr[0, 0] = [1.2, 0.6, 0.3, 2]      # /2 -> 0.6, 0.3, 0.15 -> 153, 77 (76.5 rounds up), 38
    fr[0, 1] = [2.0, 2.0, 2.0, 2]      # -> 255,255,255
    fr[1, 2] = [-1.0, np.nan, 9.0, 2]  # clamp: 0, NaN -> 0, 255
    d = oracle.display(fr, 2, java_bytes=False)
    assert d.shape == (2, 3, 3)
    assert d[1, 0].tolist() == [153, 77, 38] and d[1, 1].tolist() == [255, 255, 255] and d[0, 2].tolist() == [0, 0, 255]   # row 0 of FRAME is the bottom row
    j = oracle.display(fr, 2, java_bytes=True)

    def java(r, g, b):          # int arithmetic of dispatch.java:819-822 with signed bytes
        sb = lambda v: v - 256 if v >= 128 else v
        pix = ((sb(r) << 16) + (sb(g) << 8) + sb(b)) & 0xFFFFFFFF
        return [(pix >> 16) & 255, (pix >> 8) & 255, pix & 255]
    assert j[1, 0].tolist() == java(153, 77, 38) == [153, 77, 38]
    assert j[1, 1].tolist() == java(255, 255, 255) == [254, 254, 255]      # each channel >= 128 borrows from the one above
    assert j[0, 2].tolist() == java(0, 0, 255) == [255, 255, 255]          # blue >= 128 borrows through green into red


def test_n2_direct_diffuse_known_answers(pt, oracle):
    """directDiffuse (frag.glsl:661): col = Ka + 0.2 Kd + Kd * N.y + Ke with the UNFLIPPED normal; bgCol on a miss"""
    W, H = 8, 6
    kd, ke = (0.5, 0.25, 0.75), (0.1, 0.2, 0.3)
    for n, ny in (("0 0 -1", 0.0), ("0 1 0", 1.0), ("0 -1 0", -1.0)):
        b, sky = _scene(pt, W, H, BIG_TRI.format(z=5, z2=500, n=n), mats=[dict(Kd=kd, Ke=ke, Pr=1)], sky=(255, 255, 255))
        b[4][9] = 0.0
        fr = _render(oracle, b, sky, W, H)
        exp = np.float32(0.2) * np.array(kd, np.float32) + np.array(kd, np.float32) * np.float32(ny) + np.array(ke, np.float32)
        assert np.allclose(fr[..., :3], exp, rtol=1e-6, atol=1e-7), n
    b, sky = _scene(pt, W, H, mats=[{}], sky=(51, 102, 204))
    b[4][9] = 0.0
    fr = _render(oracle, b, sky, W, H)
    assert np.allclose(fr[..., :3], np.array([51, 102, 204], np.float32) / 255, rtol=3e-7)


def test_unorm8_reciprocal_form(tmp_path):
    """The device converts a texel byte with b * RN(1/255) and one Newton step (two fmaf) instead of the division the oracle and the GL specification
    write (pt_device.hpp unorm8): bit-equal to the IEEE quotient for all 256 bytes, checked here with the host's fmaf."""
    import subprocess
    src = tmp_path / "u.c"
    src.write_text('#include <math.h>\n#include <stdio.h>\n#include <string.h>\nint main(void){int bad=0;const float r=0x1.010102p-8f;'
                   'for(int b=0;b<256;b++){float fb=(float)b,d=fb/255.0f,q=fb*r,q2=fmaf(fmaf(-255.0f,q,fb),r,q);if(memcmp(&d,&q2,4))bad++;}'
                   'float one=1.0f/255.0f;printf("%d %d\\n",bad,memcmp(&one,&r,4)!=0);return 0;}\n')
    exe = tmp_path / "u"
    subprocess.check_call(["gcc", "-O0", "-ffp-contract=off", str(src), "-o", str(exe), "-lm"])
    assert subprocess.check_output([str(exe)], text=True).split() == ["0", "0"]


# ------------------------------------------------------------------------------------------ object order (round 6; statistics of the oracle, no rendered value)
def _duplicate_wall_workload(pt, order):
    """two `o` groups that hold the SAME wall (coplanar identical triangles, coordinates that are not dyadic so that box and triangle distances round
    apart); group B also holds a small quad nearer to the camera, so B's root box is met first"""
    scenes = pt.scenes
    sc = scenes._new_scene(); scenes._cornell_materials(sc)
    o = scenes.Obj()
    wall = ((-1.1, 0.05, 0.93), (0.9, 0.05, 0.93), (0.9, 1.9, 0.93), (-1.1, 1.9, 0.93), (0, 0, -1))
    for g in order:
        o.group("wall" + g); o.usemtl("white" if g == "A" else "red"); o.quad(*wall)
        if g == "B":
            o.quad((0.9, 0, 0), (1, 0, 0), (1, 0.1, 0), (0.9, 0.1, 0), (0, 0, -1))
    sc.addObjectText(o.text(), 0, parentDirectory="")
    return scenes._finish("dup", sc, 96, 54, scenes.CORNELL_CAM, scenes.CORNELL_ROT, (10, 20, 30), 8, 4)


def test_object_order_whatif(pt, oracle):
    """rayScene's object loop (frag.glsl:563-577) restated in other orders on the same rays (frag_oracle.cpp: xObjectLoop; profiles/r06_a_object_order_whatif.txt).
    Pins the two findings the kernel's order rests on: (1) index order over the root boxes a ray meets reproduces the reference's hit records and is what the
    oracle's node counter minus the pops of missed roots says; (2) `nearest root first, <= for a BVH of lower index than the winner` is NOT the reference on
    exact ties: a box distance rounds above its own triangle's t, so the lower-index twin is pruned — a margin on the pruning bound or a bounding pre-pass is."""
    def diffs(wl, mode):
        oracle.set_whatif(mode)
        try:
            _, cnt = oracle.render(oracle.Scene.from_workload(wl), wl.W, wl.H, 1, 1234)
        finally:
            oracle.set_whatif(0)
        c = dict(zip(oracle.COUNTERS, [int(v) for v in cnt]))
        return c
    c6 = pt.scenes.build("C6", 96, 54)
    base = diffs(c6, 9)
    assert base["xdiff"] == 0 and base["xnodes"] < base["nodes"] and base["xtris"] == base["tritests"]
    near = diffs(c6, 1)
    assert near["xdiff"] == 0 and 0.90 * base["xnodes"] < near["xnodes"] < base["xnodes"]        # the whole benefit of the other order on C6: 6 % of the visits
    ab, ba = _duplicate_wall_workload(pt, "AB"), _duplicate_wall_workload(pt, "BA")
    assert diffs(ab, 1)["xdiff"] > 100                     # the rule as proposed loses the lower-index twin on a few per cent of the rays ...
    assert diffs(ba, 1)["xdiff"] == 0
    for mode in (2, 5, 9):                                # ... the margin and the pre-pass forms do not
        assert diffs(ab, mode)["xdiff"] == 0 and diffs(ba, mode)["xdiff"] == 0
