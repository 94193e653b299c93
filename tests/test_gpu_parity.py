"""-m gpu: the HIP path (through the C ABI) against the oracle on the same seeded inputs.

Bar: the numeric contract makes the two bit-identical; the tests assert bit
equality and additionally the north_star tolerance (per-pixel RMSE <= 1e-3).
"""
import numpy as np
import pytest

from conftest import rmse

pytestmark = pytest.mark.gpu
TOL_RMSE = 1e-3          # BASELINE.json north_star: per-pixel RMSE <= 1e-3 vs reference


def seeds_for(pt, first, n):
    return [pt.scenes.frame_seed(f) for f in range(first, first + n)]


def render_both(pt, oracle, renderer_mod, wl, n_frames, first=1, count_stats=True, expect_asm=False, **opts):
    W, H = wl.W, wl.H
    seeds = seeds_for(pt, first, n_frames)
    r = renderer_mod.Renderer(W, H)
    for k, v in opts.items():
        r.set_option(k, v)
    r.set_option("count_stats", 1 if count_stats else 0)
    r.load_workload(wl)
    r.reset_frame()
    r.reset_counters()
    r.render_batch(first, seeds)
    got = r.read_frame()
    cnt = r.counters()
    if count_stats:
        # the statistics pass runs the counting variants (the compiled intersect kernel, k_shade<STATS>); the SHIPPED kernels — the hand-written
        # intersect kernel wherever the scene is one it takes — render the same frames again in the same context and must give the same bits
        r.set_option("count_stats", 0)
        r.reset_frame()
        r.render_batch(first, seeds)
        shipped = r.read_frame()
        same = (shipped == got) | (np.isnan(shipped) & np.isnan(got))
        assert same.all(), f"shipped kernels differ from the counting variants in {int((~same).sum())} floats"
        if expect_asm:
            r.set_option("query_asm_launches_above", 0)          # raises unless the hand-written intersect kernel rendered the second pass
    r.close()
    sc = oracle.Scene.from_workload(wl)
    ref, ocnt = oracle.render_frames(sc, W, H, first, n_frames, seeds, nthreads=8)
    return got, ref, cnt, dict(zip(oracle.COUNTERS, [int(x) for x in ocnt]))


def assert_same(got, ref, cnt=None, ocnt=None):
    assert rmse(got, ref) <= TOL_RMSE
    diff = ~((got == ref) | (np.isnan(got) & np.isnan(ref)))
    assert diff.sum() == 0, f"{diff.sum()} of {got.size} floats differ; max abs {np.nanmax(np.abs(got - ref))}"
    if cnt is not None:
        for k in ("segments", "nodes", "tritests", "hitupd", "samples", "boxtests"):
            assert cnt[k] == ocnt[k], (k, cnt[k], ocnt[k])


@pytest.mark.parametrize("fn,lo,hi", [("sin", -20, 20), ("cos", -20, 20), ("log", 1e-10, 10), ("exp", -100, 88), ("asin", -1.1, 1.1)])
def test_math_contract_bits(oracle, renderer_mod, fn, lo, hi):
    rs = np.random.RandomState(1)
    x = rs.uniform(lo, hi, size=1 << 18).astype(np.float32)
    x[:8] = [0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 1e-40]
    r = renderer_mod.Renderer(64, 64)
    got = r.debug_math(fn, x)
    r.close()
    ref = oracle.math(fn, x)
    assert np.array_equal(got.view(np.uint32)[~np.isnan(ref)], ref.view(np.uint32)[~np.isnan(ref)])
    assert np.array_equal(np.isnan(got), np.isnan(ref))


def test_math_contract_atan2_and_unit_randoms(oracle, renderer_mod):
    rs = np.random.RandomState(2)
    x = rs.normal(size=1 << 18).astype(np.float32); y = rs.normal(size=1 << 18).astype(np.float32)
    x[:6] = [0, 0, 0, 1, -1, np.nan]; y[:6] = [0, 1, -1, 0, 0, 1]
    r = renderer_mod.Renderer(64, 64)
    got = r.debug_math("atan2", x, y)
    # every value random() can return near the ends (log argument of Box-Muller)
    u = (np.arange(1 << 16, dtype=np.float64) / 2.0 ** 32).astype(np.float32)
    gl = r.debug_math("log", u)
    r.close()
    ref = oracle.math("atan2", x, y)
    assert np.array_equal(np.isnan(got), np.isnan(ref)) and np.array_equal(got[~np.isnan(ref)], ref[~np.isnan(ref)])
    assert np.array_equal(gl.view(np.uint32), oracle.math("log", u).view(np.uint32))


def test_k1_rng_vectors_on_the_device(renderer_mod):
    """the RNG known-answer vectors of SURVEY.md §8(c) (frag.glsl:686-694) through the DEVICE's NextRandom / random(), not only the oracle's"""
    from test_oracle_kat import RNG_KAT
    r = renderer_mod.Renderer(64, 64)
    for start, rows in RNG_KAT.items():
        st = np.array([start], dtype=np.uint32)
        for exp_state, exp_res, exp_bits in rows:
            res = r.debug_math("rng_result", st.view(np.float32)).view(np.uint32)
            rnd = r.debug_math("rng_random", st.view(np.float32)).view(np.uint32)
            st = r.debug_math("rng_state", st.view(np.float32)).view(np.uint32).copy()
            assert (int(st[0]), int(res[0]), int(rnd[0])) == (exp_state, exp_res, exp_bits)
    r.close()


def test_unorm8_on_the_device(renderer_mod):
    """byte / 255.0f as the texture samplers evaluate it (reciprocal + one Newton step, pt_device.hpp unorm8) against the IEEE quotient, all 256 bytes"""
    b = np.arange(256, dtype=np.float32)
    r = renderer_mod.Renderer(32, 32)
    got = r.debug_math("unorm8", b)
    r.close()
    assert np.array_equal(got.view(np.uint32), (b / np.float32(255.0)).view(np.uint32))


def asm_taken(r):
    """does this context's scene run on the hand-written intersect kernel (pt_extend_gfx950.s)?"""
    try:
        r.set_option("query_asm_eligible", 0)
        return True
    except Exception:
        return False


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("name,W,H", [("C2", 96, 54), ("C3", 96, 54), ("C1", 64, 64)])
def test_intersect_parity(pt, oracle, renderer_mod, name, W, H, mode):
    wl = pt.scenes.build(name, W, H)
    rs = np.random.RandomState(3)
    n = 4096
    o = (np.array(wl.buffers[0]) + rs.normal(scale=0.3, size=(n, 3))).astype(np.float32)
    d = rs.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[:4] = [[1, 0, 0], [0, -1, 0], [0, 0, 1], [0, 0, 0]]
    r = renderer_mod.Renderer(W, H)
    r.load_workload(wl)
    r.set_option("extend_mode", mode)          # 0: one block per 256 rays; 1: persistent, compiled; 2: persistent, hand-written
    tuv, prim = r.debug_intersect(o, d)
    if mode == 2:
        assert asm_taken(r)
        r.set_option("query_asm_launches_above", 0)
    r.close()
    sc = oracle.Scene.from_workload(wl)
    for i in range(n):
        code, out = oracle.ray_scene(sc, o[i], d[i])
        if code < 0:
            assert prim[i] == -1 or not (tuv[i, 0] < 1e25), (i, prim[i], tuv[i])
        else:
            typ, pid = code >> 24, code & 0xFFFFFF
            exp = pid if typ == 1 else (0x40000000 | pid)
            assert prim[i] == exp and tuv[i, 0] == out[0], (i, prim[i], exp, tuv[i, 0], out[0])


def _raybox_np(o, invd, mn, mx, ieee):
    """rayBox (frag.glsl:408-419) on arrays, with IEEE minNum/maxNum (what v_min_f32 / v_max_f32 and every GPU's GLSL min/max do) or
    with the GLSL specification's min(x,y) = (y < x) ? y : x, max(x,y) = (x < y) ? y : x"""
    with np.errstate(all="ignore"):
        tmin = ((mn - o) * invd).astype(np.float32); tmax = ((mx - o) * invd).astype(np.float32)
        if ieee:
            mn2, mx2 = np.fmin, np.fmax
        else:
            mn2 = lambda x, y: np.where(y < x, y, x)          # noqa: E731
            mx2 = lambda x, y: np.where(x < y, y, x)          # noqa: E731
        t1, t2 = mn2(tmin, tmax), mx2(tmin, tmax)
        tnear = mx2(mx2(t1[..., 0], t1[..., 1]), t1[..., 2]); tfar = mn2(mn2(t2[..., 0], t2[..., 1]), t2[..., 2])
        return np.where((tfar >= tnear) & (tfar > 0), np.where(tnear > 0, tnear, 0), np.float32(1e30))


def test_nan_slab_rays(pt, oracle, renderer_mod):
    """rayBox with a 0 * inf slab (frag.glsl:409-417): a ray that starts exactly ON a box plane and runs exactly parallel to it.  The two
    readings of min/max differ only when BOTH slab distances are NaN — a flat box (an axis-aligned wall) with the ray in its plane: minNum
    ignores the slab, the specification's formula misses the box.  Both sides of the parity pair use minNum; this test
    runs such rays through both, and checks that the set really contains boxes on which the two readings differ."""
    wl = pt.scenes.build("C2", 96, 54)
    boxes = np.array(wl.buffers[10], np.float32).reshape(-1, 8)
    mn, mx = boxes[:, 0:3], boxes[:, 3:6]
    O, D = [], []
    for k in range(len(boxes)):
        c = (mn[k] + mx[k]) * np.float32(0.5)
        for a in range(3):
            for plane in (mn[k, a], mx[k, a]):
                for b in range(3):
                    if b == a:
                        continue
                    for sgn in (1.0, -1.0):
                        o = c.copy(); o[a] = plane; o[b] -= np.float32(sgn * 3.0)
                        d = np.zeros(3, np.float32); d[b] = sgn
                        d[a] = 0.0 if sgn > 0 else -0.0                      # 1/d = +inf and -inf
                        O.append(o); D.append(d)
    O, D = np.array(O, np.float32), np.array(D, np.float32)
    with np.errstate(all="ignore"):
        invd = (np.float32(1.0) / D).astype(np.float32)
    a_ieee = _raybox_np(O[:, None, :], invd[:, None, :], mn[None], mx[None], True)
    a_spec = _raybox_np(O[:, None, :], invd[:, None, :], mn[None], mx[None], False)
    assert (a_ieee != a_spec).sum() > 100, "the ray set must contain boxes on which minNum and the specification's min/max differ"
    r = renderer_mod.Renderer(96, 54)
    r.load_workload(wl)
    tuv, prim = r.debug_intersect(O, D)         # the hand-written kernel: these rays are the ones that switch it to the min/max form of rayBox
    r.set_option("query_asm_launches_above", 0)
    r.set_option("extend_mode", 1)
    tuv1, prim1 = r.debug_intersect(O, D)
    r.close()
    assert np.array_equal(prim, prim1) and np.array_equal(tuv.view(np.uint32), tuv1.view(np.uint32))
    sc = oracle.Scene.from_workload(wl)
    hits = 0
    for i in range(len(O)):
        code, out = oracle.ray_scene(sc, O[i], D[i])
        if code < 0:
            assert prim[i] == -1 or not (tuv[i, 0] < 1e25), (i, prim[i], tuv[i])
        else:
            hits += 1
            assert prim[i] == (code & 0xFFFFFF) and tuv[i, 0] == out[0], (i, prim[i], code, tuv[i, 0], out[0])
    assert hits > 50
    # and through the whole path with traversal counters: a camera whose primary rays are axis-parallel in one component
    wl2 = wl.with_params(BLUR=0.0, AUTO_FOCUS=0)
    b = dict(wl2.buffers); b[0] = np.array([0.0, float(mx[0, 1]), -3.0], np.float32); b[1] = np.zeros(3, np.float32)
    wl2 = pt.scenes.Workload("C2_grazing", wl2.W, wl2.H, b, wl2.sky, wl2.sample_res, wl2.max_bounces, wl2.info)
    got, ref, cnt, ocnt = render_both(pt, oracle, renderer_mod, wl2, 2)
    assert_same(got, ref, cnt, ocnt)
    got, ref, _, _ = render_both(pt, oracle, renderer_mod, wl2, 2, count_stats=False)     # ... and on the hand-written kernel (statistics run on the compiled one)
    assert_same(got, ref)


def mixed_regular_and_axis_parallel_rays(wl, n, seed):
    rs = np.random.RandomState(seed)
    o = (np.array(wl.buffers[0]) + rs.normal(scale=0.4, size=(n, 3))).astype(np.float32)
    d = rs.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    e = np.asarray(wl.buffers[7]); ne = int(e[0])
    if ne:                                                 # a quarter of the rays start in or near an ellipsoid (inside: the negative near root), half of those aim at one
        c = e[1:1 + 3 * ne].reshape(ne, 3); rad = e[1 + 9 * ne:1 + 10 * ne]
        k = np.arange(0, n, 4)
        which = rs.randint(0, ne, size=k.size)
        o[k] = (c[which] + rs.normal(size=(k.size, 3)) * rad[which, None] * rs.choice([0.3, 1.0, 2.5], size=(k.size, 1))).astype(np.float32)
        k2 = np.arange(1, n, 8)
        to = c[rs.randint(0, ne, size=k2.size)] + rs.normal(scale=0.05, size=(k2.size, 3)) - o[k2]
        d[k2] = (to / np.linalg.norm(to, axis=1, keepdims=True)).astype(np.float32)
    k = rs.randint(0, n, size=n // 16)                     # every wave gets a few rays with a zero / negative-zero / NaN / infinite component
    d[k, rs.randint(0, 3, size=k.size)] = rs.choice(np.array([0.0, -0.0, np.nan, np.inf, 1e-42], np.float32), size=k.size)
    k = rs.randint(0, n, size=n // 64)
    o[k, rs.randint(0, 3, size=k.size)] = rs.choice(np.array([np.inf, np.nan, 1.0, -1.0, 0.0], np.float32), size=k.size)
    return o, d


@pytest.mark.parametrize("name,W,H,kw,opts", [("C2", 96, 54, {}, {}), ("C2", 96, 54, {}, {"asm_loop": 1}), ("C3", 96, 54, {}, {}), ("C3", 96, 54, {}, {"asm_loop": 0}),
                                              ("C4", 64, 36, {}, {"asm_loop": 0}), ("C3", 96, 54, {}, {"extend_cache_bytes": 0, "refill_min": 1}),
                                              ("C3", 96, 54, {}, {"extend_cache_bytes": 60000, "refill_min": 64, "none_min": 1}),
                                              ("C5", 64, 36, {"subdiv": 2}, {"inner_keep_eighths": 0}), ("C4", 64, 36, {}, {}), ("C5", 64, 36, {}, {"stack_mode": 1}),
                                              ("C3", 96, 54, {}, {"asm_tpb": 1024}), ("C2", 96, 54, {}, {"asm_tpb": 1024}), ("C4", 64, 36, {}, {"asm_tpb": 1024}),
                                              ("C5", 64, 36, {}, {"asm_tpb": 1024, "stack_mode": 1, "asm_loop": 0}),
                                              ("C3", 96, 54, {}, {"asm_tpb": 1024, "extend_blocks_per_cu": 4, "extend_cache_bytes": 114688, "refill_min": 1}),
                                              # ellipsoids (tested when the hit records leave) and more than 8 BVHs (root records in LDS, tested when a BVH's turn comes)
                                              ("C1", 64, 64, {}, {}), ("C1", 64, 64, {}, {"asm_loop": 1, "refill_min": 1}), ("C1rot", 64, 64, {}, {"asm_tpb": 1024}),
                                              ("C6", 64, 36, {}, {}), ("C6", 64, 36, {}, {"asm_loop": 0}), ("C6", 64, 36, {}, {"asm_tpb": 1024, "none_min": 1}),
                                              # node records of the other layout than the automatic choice (80-B sign-ordered for trees that fit the caches, else 64-B + the min/max step)
                                              ("C3", 96, 54, {}, {"asm_node_layout": 1}), ("C3", 96, 54, {}, {"asm_node_layout": 1, "asm_loop": 0, "asm_tpb": 1024}), ("C4", 64, 36, {}, {"asm_node_layout": 0}),
                                              ("C6", 64, 36, {}, {"asm_node_layout": 1}), ("C6", 64, 36, {}, {"asm_node_layout": 0, "asm_loop": 0}), ("C2", 96, 54, {}, {"asm_node_layout": 1}),
                                              ("C6", 64, 36, {"groups": 11, "nu": 8, "nv": 8}, {"refill_min": 64}), ("C6", 64, 36, {"groups": 9, "nu": 6, "nv": 6}, {"asm_loop": 0, "none_min": 32}),
                                              # the per-ray cull of the object loop (round 5): off; more BVHs than mask bits — groups of 2, 4 and 16 BVHs per bit
                                              ("C6", 64, 36, {}, {"asm_root_cull": 0}), ("C6", 64, 36, {"groups": 70, "nu": 6, "nv": 6}, {}),
                                              ("C6", 64, 36, {"groups": 130, "nu": 4, "nv": 4}, {"asm_loop": 0, "refill_min": 1}), ("C6", 64, 36, {"groups": 600, "nu": 4, "nv": 4}, {"asm_tpb": 1024}),
                                              ("C6", 64, 36, {"groups": 65, "nu": 4, "nv": 4}, {"asm_node_layout": 1, "none_min": 1}),
                                              # 24-bit traversal-stack entries (16 bits + a byte in a second LDS array: what trees beyond 131071 nodes get), forced onto small trees
                                              ("C3", 96, 54, {}, {"stack_mode": 2}), ("C4", 64, 36, {}, {"stack_mode": 2, "asm_loop": 0}), ("C5", 64, 36, {}, {"stack_mode": 2, "asm_tpb": 1024}),
                                              ("C6", 64, 36, {}, {"stack_mode": 2, "refill_min": 1}),
                                              # 512-thread blocks
                                              ("C3", 96, 54, {}, {"asm_tpb": 512}), ("C4", 64, 36, {}, {"asm_tpb": 512, "asm_loop": 0}), ("C6", 64, 36, {}, {"asm_tpb": 512, "refill_min": 8})])
def test_handwritten_intersect_kernel_equals_compiled(pt, renderer_mod, name, W, H, kw, opts):
    """pt_extend_gfx950.s against the compiled kernels on 64 K rays per scene, a sixteenth of them irregular (zero, denormal, infinite, NaN
    components: the rays that take its min/max step): every hit record bit for bit.  C4 and the last case run its 18-bit-stack form;
    asm_loop picks its main loop (0 phase-voting, 1 fused trip; automatic: fused unless the whole scene sits in the LDS tile, as C2 does);
    asm_tpb its block size (1024 threads: what a context alone on its GPU launches once a launch is large enough; the last case one block
    per CU with a 112 KB tile).  C1 / C6: ellipsoids (a quarter of the rays start in or near one; C1rot: all three rotated) and 64 / 11 / 9 BVHs."""
    if name == "C1rot":
        wl = pt.scenes.build("C1", W, H)
        b = dict(wl.buffers); e = b[7].copy(); ne = int(e[0])
        e[1 + 6 * ne: 1 + 9 * ne] = [0.3, 0.2, 0.1, -0.4, 0.0, 0.25, 0.0, 1.1, 0.0]
        e[1 + 3 * ne: 1 + 6 * ne] = [1.0, 2.0, 0.5, 1.5, 1.0, 1.0, 0.7, 0.7, 2.0]
        b[7] = e
        wl = pt.scenes.Workload("C1rot", W, H, b, wl.sky, wl.sample_res, wl.max_bounces, wl.info)
    else:
        wl = pt.scenes.build(name, W, H, **kw)
    o, d = mixed_regular_and_axis_parallel_rays(wl, 1 << 16, 11)
    r = renderer_mod.Renderer(W, H)
    r.load_workload(wl)
    for k, v in opts.items():
        r.set_option(k, v)
    assert asm_taken(r)
    out = {}
    for mode in (2, 1, 0):
        r.set_option("extend_mode", mode)
        tuv, prim = r.debug_intersect(o, d)
        out[mode] = (tuv.view(np.uint32).copy(), prim.copy())
    r.set_option("query_asm_launches_above", 0)
    r.close()
    for mode in (1, 0):
        same = (out[2][1] == out[mode][1]) & (out[2][0] == out[mode][0]).all(axis=1)
        assert same.all(), (mode, int((~same).sum()), np.nonzero(~same)[0][:8], out[2][1][~same][:8], out[mode][1][~same][:8])
    assert (out[2][1] >= 0).sum() > 1000


def test_implicit_surfaces_are_accepted_and_never_hit(pt, oracle, renderer_mod):
    """scene.addImplicit (dispatch.java:1005-1011) fills binding 5; the shader loops over the implicits (frag.glsl:578-605) but rayImplicit returns 1e30 before anything
    else (:385-386), so they are never hit: a scene WITH implicits renders the image of the scene without them — here and in the oracle, which both used to refuse it"""
    W, H = 96, 54

    def build(with_implicits):
        sc = pt.hostlib.Scene()
        sc.addMaterial("default"); sc.setLastMtl("Kd", (0.8, 0.8, 0.8)); sc.setLastMtl("Pr", 1)
        sc.addMaterial("lamp"); sc.setLastMtl("Ke", (12, 12, 12))
        quad = ("o floor\nvn 0 1 0\nv -2 0 -2\nv 2 0 -2\nv 2 0 2\nv -2 0 2\nf 1//1 2//1 3//1\nf 1//1 3//1 4//1\n"
                "o lamp\nusemtl lamp\nvn 0 -1 0\nv -0.5 2 -0.5\nv 0.5 2 -0.5\nv 0.5 2 0.5\nv -0.5 2 0.5\nf 5//2 7//2 6//2\nf 5//2 8//2 7//2\n")
        sc.addObjectText(quad, 0, parentDirectory="")
        sc.addEllipsoid((0.0, 0.5, 0.0), 1.0, 0.0, 0.5, 0)
        if with_implicits:
            sc.addImplicit(2, (0.0, 0.6, 0.0), (0.5, 0.5, 0.5), (0.0, 0.0, 0.0), 1)
            sc.addImplicit(0, (0.3, 0.2, -0.4), (1.0, 2.0, 1.0), (0.2, 0.1, 0.0), 0)
        return pt.scenes._finish("implicits", sc, W, H, (0.0, 1.0, -3.0), (0.0, 0.0, 0.0), (150, 180, 230), 8, 4)
    with_i, without = build(True), build(False)
    assert int(with_i.buffers[5][0]) == 2 and int(without.buffers[5][0]) == 0
    got, ref, cnt, ocnt = render_both(pt, oracle, renderer_mod, with_i, 3)
    assert_same(got, ref, cnt, ocnt)
    plain, ref0, _, _ = render_both(pt, oracle, renderer_mod, without, 3)
    assert np.array_equal(got, plain, equal_nan=True) and np.array_equal(ref, ref0, equal_nan=True)


@pytest.mark.parametrize("perturb", ["children_stick_out", "root_box_shrunk_to_nothing", "inverted_root_box", "inverted_child_box"])
def test_root_cull_only_where_the_boxes_promise_it(pt, renderer_mod, perturb):
    """The per-ray cull of the object loop skips a BVH whose ROOT box the ray misses; that is rayBVH's own result only if the root's child boxes lie inside an
    ordered root box (the reference's builder guarantees it, the C ABI takes any buffer).  Root boxes made smaller than their children, shrunk to a point far
    away, or inverted: the hand-written kernel must still equal the compiled kernels (which test every root box in turn) bit for bit.
    inverted_child_box (round-5 advisor): the root box covers only the lower half of its triangles and both children are stored (max, min) — each plane of a child
    passes a one-sided test against the root (min >= root.min, max <= root.max) while the slab rayBox makes of it, [max, min], sticks out of the root; a ray with
    closest_t still 1e30 that misses the root is traversed by the reference (1e30 > 1e30 is false) and meets the triangles of the upper half."""
    W, H = 64, 36
    wl = pt.scenes.build("C6", W, H, groups=24, nu=6, nv=6)
    b = dict(wl.buffers)
    data = np.array(b[10], dtype=np.float32).copy().reshape(-1, 8)
    roots = np.asarray(b[13])[1:1 + int(b[13][0])]
    for k, r in enumerate(roots[2::3]):                       # every third torus
        lo, hi = data[r, 0:3].copy(), data[r, 3:6].copy()
        if perturb == "children_stick_out":
            mid = 0.5 * (lo + hi); data[r, 0:3] = mid - 0.25 * (hi - lo); data[r, 3:6] = mid + 0.25 * (hi - lo)
        elif perturb == "root_box_shrunk_to_nothing":
            data[r, 0:3] = 50.0 + k; data[r, 3:6] = 50.0 + k
        elif perturb == "inverted_child_box":
            tree = np.asarray(b[11]).reshape(-1, 3)
            data[r, 4] = 0.5 * (lo[1] + hi[1])                # the root: lower half in y
            for ch in (int(tree[r, 1]), int(tree[r, 2])):
                cl, chh = data[ch, 0:3].copy(), data[ch, 3:6].copy()
                data[ch, 0:3] = chh; data[ch, 3:6] = cl
        else:
            data[r, 0:3] = hi; data[r, 3:6] = lo
    b[10] = data.reshape(-1)
    wl = pt.scenes.Workload("C6p", W, H, b, wl.sky, wl.sample_res, wl.max_bounces, wl.info)
    o, d = mixed_regular_and_axis_parallel_rays(wl, 1 << 16, 12)
    if perturb == "inverted_child_box":                       # a quarter of the rays start inside the room and leave through its open side: no wall behind the tori, closest_t stays 1e30
        rs = np.random.RandomState(5); n4 = o.shape[0] // 4
        o[:n4] = rs.uniform((-0.9, 0.1, -0.9), (0.9, 1.9, 0.9), size=(n4, 3)).astype(np.float32)
        dd = rs.normal(size=(n4, 3)); dd[:, 2] = -np.abs(dd[:, 2]) - 0.5
        d[:n4] = (dd / np.linalg.norm(dd, axis=1, keepdims=True)).astype(np.float32)
    r = renderer_mod.Renderer(W, H)
    r.load_workload(wl)
    if perturb in ("inverted_root_box", "inverted_child_box"):
        r.set_option("asm_node_layout", 1)                    # (an inverted box is what the 80-B sign-ordered records cannot hold: such scenes run the compiled kernel; the 64-B layout's min/max step takes them)
    assert asm_taken(r)
    out = {}
    for mode in (2, 1):
        r.set_option("extend_mode", mode)
        tuv, prim = r.debug_intersect(o, d)
        out[mode] = (tuv.view(np.uint32).copy(), prim.copy())
    r.set_option("query_asm_launches_above", 0)
    r.close()
    same = (out[2][1] == out[1][1]) & (out[2][0] == out[1][0]).all(axis=1)
    assert same.all(), (int((~same).sum()), np.nonzero(~same)[0][:8])
    assert (out[2][1] >= 0).sum() > 1000


@pytest.mark.parametrize("name,W,H,frames,kw,opts", [("C2", 96, 54, 3, {}, {}), ("C3", 128, 72, 3, {}, {}), ("C3", 128, 72, 4, {}, {"path_slots": 2048}),
                                                     ("C3", 128, 72, 2, {}, {"path_slots": 1 << 16, "refill_min": 8}), ("C5", 64, 36, 2, {"subdiv": 2}, {}),
                                                     ("C4", 64, 36, 2, {}, {}), ("T1", 96, 54, 2, {}, {}), ("C3", 128, 72, 3, {}, {"asm_loop": 0}),
                                                     ("C2", 96, 54, 2, {}, {"asm_loop": 1, "path_slots": 1024}),
                                                     ("C3", 128, 72, 3, {}, {"asm_tpb": 1024}), ("C3", 128, 72, 3, {}, {"asm_tpb": 1024, "path_slots": 2048}),
                                                     ("C4", 64, 36, 2, {}, {"asm_tpb": 1024}), ("C5", 64, 36, 2, {"subdiv": 2}, {"asm_tpb": 1024}),
                                                     ("C1", 96, 96, 3, {}, {}), ("C6", 96, 54, 2, {}, {}), ("C6", 96, 54, 2, {}, {"asm_tpb": 1024, "path_slots": 2048}),
                                                     ("C3", 128, 72, 3, {}, {"stack_mode": 2}), ("C4", 64, 36, 2, {}, {"stack_mode": 2, "path_slots": 2048}),
                                                     ("C6", 96, 54, 2, {"groups": 70, "nu": 6, "nv": 6}, {}), ("C6", 96, 54, 2, {"groups": 200, "nu": 4, "nv": 4}, {"path_slots": 2048})])
def test_render_parity_handwritten_kernel(pt, oracle, renderer_mod, name, W, H, frames, kw, opts):
    """whole renders on the hand-written intersect kernel (statistics off: the counting variant is the compiled kernel) against the oracle,
    including the small pool that hands over to the device-packed tail queue"""
    wl = pt.scenes.build(name, W, H, **kw)
    seeds = seeds_for(pt, 1, frames)
    r = renderer_mod.Renderer(W, H)
    for k, v in opts.items():
        r.set_option(k, v)
    r.load_workload(wl)
    r.reset_frame()
    r.render_batch(1, seeds)
    got = r.read_frame()
    assert asm_taken(r)
    r.set_option("query_asm_launches_above", 3)
    r.close()
    ref, _ = oracle.render_frames(oracle.Scene.from_workload(wl), W, H, 1, frames, seeds, nthreads=8)
    assert_same(got, ref)


@pytest.mark.parametrize("groups", [64, 65])
@pytest.mark.parametrize("tpb", [256, 1024])
def test_group_cull_boundaries_at_full_size(pt, oracle, renderer_mod, groups, tpb):
    """The one abort this repository ever saw on a committed-next kernel (gpurun r05b, DESIGN.md section 2.2: the first FULL-SIZE many-BVH render, C6 at 1920x1080 on the
    1024-thread blocks a context alone on its GPU takes) was found by no small case: the per-ray group cull at its boundaries — 64 groups (one BVH per mask bit, every
    bit used) and 65 (two BVHs per bit, 33 bits) — on both block sizes AT 1920x1080, where a launch has its 512 large blocks, the LDS allocation is the production
    one (32 KB tile + root records + group boxes + 16 stack levels x 1024 lanes) and 16 M rays per frame reach the deep ends of the traversal stacks.
    One frame each, against the oracle on a pixel lattice."""
    W, H = 1920, 1080
    wl = pt.scenes.build("C6", W, H, groups=groups, nu=10, nv=10)
    seeds = seeds_for(pt, 1, 1)
    r = renderer_mod.Renderer(W, H)
    r.set_option("asm_tpb", tpb)
    r.load_workload(wl); r.reset_frame()
    r.render_batch(1, seeds)
    got = r.read_frame().copy()
    assert asm_taken(r)
    r.close()
    ref = np.zeros((H, W, 4), np.float32)
    oracle.render(oracle.Scene.from_workload(wl), W, H, 1, seeds[0], ref, nthreads=8, xs=32, ys=27)
    assert np.array_equal(got[::27, ::32].view(np.uint32), ref[::27, ::32].view(np.uint32)) or np.array_equal(got[::27, ::32], ref[::27, ::32], equal_nan=True)
    assert np.all(got[..., 3] == 1.0)


@pytest.mark.parametrize("name,frames,xs,ys,streams,bound", [("C1", 1, 1, 1, 1, 1e-3), ("C2", 8, 8, 9, 1, 6e-3), ("C2", 512, 16, 18, 2, 1e-3), ("C3", 32, 24, 27, 2, 1e-3),
                                                             ("C4", 128, 48, 54, 2, 1e-3), ("C5", 512, 96, 108, 2, 1e-3)])
def test_fast_contract_rmse(pt, oracle, renderer_mod, name, frames, xs, ys, streams, bound):
    """The opt-in relaxed numeric contract (pt_set_option numeric_contract = 1: hardware rcp / rsq / sqrt / log / cos; RNG, draw counts and
    branches as written) at BASELINE.json's image sizes and sample counts — C1 (one frame of 4 spp, whole image), C2 (64 spp), C3 (256 spp),
    C4 (1024 spp), C5 (4096 spp) — against the (exact) oracle on a pixel lattice: per-pixel RMSE <= 1e-3, north_star's tolerance.
    ONE configuration cannot meet it and says so: C2 at its 64 spp measures 4.1e-3.  Its pixels are sums of products of the materials' constants,
    so 12787 of the 12800 lattice pixels are bit-identical to the oracle and the whole error sits in 13 pixels where ONE of the 64 x 8 path
    segments landed on the other side of a triangle edge (a 15x emitter: +-0.17 per path) — what any evaluation that is not bit-exact does
    about 8 times per million segments.  That error falls as 1/sqrt(spp): the same scene at 4096 spp is the third case (<= 1e-3).
    The results are NOT bit-identical (asserted too: a fast mode that changed nothing would not be one)."""
    cfg = pt.scenes.CONFIGS[name]
    W, H = cfg["W"], cfg["H"]
    wl = pt.scenes.build(name, W, H)
    seeds = seeds_for(pt, 1, frames)
    r = renderer_mod.Renderer(W, H, devices=[0] * streams) if streams > 1 else renderer_mod.Renderer(W, H)
    r.set_option("numeric_contract", 1)
    r.load_workload(wl); r.reset_frame()
    for k in range(0, frames, 32):
        r.render_batch_async(1 + k, seeds[k:k + 32])
    got = r.read_frame().copy()
    r.close()
    assert np.all(got[..., 3] == frames)
    sc = oracle.Scene.from_workload(wl)
    ref = np.zeros((H, W, 4), np.float32)
    for i, sd in enumerate(seeds):
        oracle.render(sc, W, H, 1 + i, sd, ref, nthreads=16, xs=xs, ys=ys)
    a, b = got[::ys, ::xs], ref[::ys, ::xs]
    e = rmse(a, b)
    differing = int((a.view(np.uint32) != b.view(np.uint32)).any(axis=-1).sum())
    print(f"{name}: relaxed contract, {frames} frames, lattice {a.shape[1]}x{a.shape[0]}: RMSE {e:.3e}, {differing} of {a.shape[0] * a.shape[1]} lattice pixels differ in some bit")
    assert e <= bound, e
    assert differing > 0


def test_fast_contract_is_opt_in_and_per_stream(pt, oracle, renderer_mod):
    """the default contract is the exact one; switching takes effect with the next frame stream and back again"""
    wl = pt.scenes.build("C3", 96, 54)
    seeds = seeds_for(pt, 1, 2)
    ref, _ = oracle.render_frames(oracle.Scene.from_workload(wl), 96, 54, 1, 2, seeds, nthreads=8)
    r = renderer_mod.Renderer(96, 54)
    r.load_workload(wl)
    r.reset_frame(); r.render_batch(1, seeds); exact0 = r.read_frame().copy()
    r.set_option("numeric_contract", 1)
    r.reset_frame(); r.render_batch(1, seeds); fast = r.read_frame().copy()
    r.set_option("numeric_contract", 0)
    r.reset_frame(); r.render_batch(1, seeds); exact1 = r.read_frame().copy()
    r.close()
    assert_same(exact0, ref); assert_same(exact1, ref)
    assert not np.array_equal(fast, ref, equal_nan=True) and rmse(fast, ref) < 0.05      # 16 spp on 5 K pixels: close, a flipped path is visible


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("name,W,H,frames", [("C1", 64, 64, 2), ("C2", 96, 54, 3), ("C3", 96, 54, 3), ("C5", 64, 36, 2)])
def test_render_parity_small(pt, oracle, renderer_mod, name, W, H, frames, mode):
    wl = pt.scenes.build(name, W, H) if name != "C5" else pt.scenes.build(name, W, H, subdiv=2)
    got, ref, cnt, ocnt = render_both(pt, oracle, renderer_mod, wl, frames, extend_mode=mode)
    assert_same(got, ref, cnt, ocnt)


@pytest.mark.parametrize("tpb,cache,refill", [(256, 0, 1), (512, 8192, 16), (1024, 65536, 48), (512, 150000, 64), (64, 1024, 8), (128, 4096, 24)])
def test_render_parity_persistent_variants(pt, oracle, renderer_mod, tpb, cache, refill):
    """persistent intersect kernel: block size, LDS tile size (partial / whole BVH) and refill threshold do not change results"""
    wl = pt.scenes.build("C3", 128, 72)
    got, ref, cnt, ocnt = render_both(pt, oracle, renderer_mod, wl, 2, extend_mode=1, extend_tpb=tpb, extend_cache_bytes=cache, refill_min=refill)
    assert_same(got, ref, cnt, ocnt)


@pytest.mark.parametrize("name,W,H,stack_mode,bfs_nodes", [("C3", 96, 54, 1, 0), ("C3", 96, 54, 2, 15), ("C4", 64, 36, -1, 1000), ("C4", 64, 36, 2, 100000), ("C5", 64, 36, 1, 3)])
def test_render_parity_stack_widths_and_node_layouts(pt, oracle, renderer_mod, name, W, H, stack_mode, bfs_nodes):
    """the three widths of the traversal-stack entries (16-bit, 16 + 2 register bits, 32-bit) and the breadth-first / depth-first record
    layouts are pure re-encodings: same bits, same traversal counters (C4's 100 k-triangle tree takes the 18-bit form by itself)"""
    wl = pt.scenes.build(name, W, H)
    got, ref, cnt, ocnt = render_both(pt, oracle, renderer_mod, wl, 2, stack_mode=stack_mode, bfs_nodes=bfs_nodes)
    assert_same(got, ref, cnt, ocnt)


@pytest.mark.parametrize("slots,extend_mode", [(2048, 1), (256, 1), (9216, 0), (1 << 16, 1)])
def test_render_parity_small_pool_and_packed_tail(pt, oracle, renderer_mod, slots, extend_mode):
    """pool smaller than (or larger than) the job count: regeneration, job pulling, the hand-over to the device-packed tail queue"""
    wl = pt.scenes.build("C3", 128, 72)
    got, ref, cnt, ocnt = render_both(pt, oracle, renderer_mod, wl, 4, path_slots=slots, extend_mode=extend_mode)
    assert_same(got, ref, cnt, ocnt)


@pytest.mark.parametrize("slots,W,H", [(1 << 16, 24, 16), (2048, 40, 30), (256, 24, 16)])
def test_pool_far_larger_than_the_first_batch(pt, oracle, renderer_mod, slots, W, H):
    """a pool of which only a block or two come alive with the first batch, and frames appended to the running stream: the few live slots (and the
    revived ones) hand out every job"""
    wl = pt.scenes.build("C3", W, H)
    seeds = seeds_for(pt, 1, 6)
    r = renderer_mod.Renderer(W, H)
    r.set_option("path_slots", slots)
    r.load_workload(wl); r.reset_frame()
    r.render_batch_async(1, seeds[:1])
    r.render_batch_async(2, seeds[1:4])         # appended to the running stream
    r.render_batch_async(5, seeds[4:])
    got = r.read_frame()
    r.close()
    ref, _ = oracle.render_frames(oracle.Scene.from_workload(wl), W, H, 1, 6, seeds, nthreads=8)
    assert_same(got, ref)


def test_batch_equals_frame_at_a_time_and_frame_counter(pt, oracle, renderer_mod):
    """K7: FRAME.rgb = sum of frames, a = N, frame 1 overwrites (frag.glsl:924-933)"""
    wl = pt.scenes.build("C2", 64, 36)
    seeds = seeds_for(pt, 1, 4)
    r = renderer_mod.Renderer(64, 36)
    r.load_workload(wl)
    r.reset_frame()
    r.render_batch(1, seeds)
    a = r.read_frame().copy()
    r.reset_frame()
    for i, s in enumerate(seeds):
        r.render(1 + i, s)
    b = r.read_frame().copy()
    r.render(1, seeds[0])            # u_frameCount == 1 overwrites the accumulator
    c = r.read_frame().copy()
    r.close()
    assert np.array_equal(a, b)
    assert np.all(a[..., 3] == 4.0) and np.all(c[..., 3] == 1.0)


@pytest.mark.parametrize("form", ["one", "two_streams", "shards", "async"])
def test_write_frame_resumes_an_accumulation(pt, oracle, renderer_mod, form):
    """FRAME (running sum + count, frag.glsl:924-933) is the path tracer's only persistent state (resumability = re-upload sum + count):
    N frames, pt_read_frame, the context destroyed, a new one, pt_write_frame, M more frames == N + M frames in one go, bit for bit and against the oracle —
    on one stream, on a two-stream context (the image is distributed over the tile shards), on single shards of three, and with the resumed frames submitted
    asynchronously into a fresh image of the ring."""
    W, H, N, M = 96, 54, 3, 2
    wl = pt.scenes.build("C3", W, H)
    seeds = seeds_for(pt, 1, N + M)
    make = (lambda: renderer_mod.Renderer(W, H, devices=[0, 0])) if form == "two_streams" else (lambda: renderer_mod.Renderer(W, H))
    r = make(); r.load_workload(wl); r.reset_frame(); r.render_batch(1, seeds)
    whole = r.read_frame().copy(); r.close()
    r = make(); r.load_workload(wl); r.reset_frame(); r.render_batch(1, seeds[:N])
    saved = r.read_frame().copy(); r.close()
    assert np.all(saved[..., 3] == N)
    if form == "shards":
        acc = np.zeros_like(whole)
        for rank in range(3):
            rr = renderer_mod.Renderer(W, H, shard_rank=rank, shard_count=3)
            rr.load_workload(wl); rr.reset_frame(); rr.write_frame(saved); rr.render_batch(N + 1, seeds[N:])
            part = rr.read_frame(); rr.close()
            assert np.all(acc[part[..., 3] > 0] == 0)
            acc += part
        got = acc
    else:
        r = make(); r.load_workload(wl)
        if form == "async":
            r.reset_frame(); r.render_batch_async(1, seeds[:1]); r.next_image()      # something else in flight in the previous image of the ring
            r.write_frame(saved)
            for k in range(M):
                r.render_batch_async(N + 1 + k, seeds[N + k:N + k + 1])
        else:
            r.reset_frame(); r.write_frame(saved); r.render_batch(N + 1, seeds[N:])
        got = r.read_frame().copy(); r.close()
    assert np.array_equal(got.view(np.uint32), whole.view(np.uint32)) and np.all(got[..., 3] == N + M)
    ref, _ = oracle.render_frames(oracle.Scene.from_workload(wl), W, H, 1, N + M, seeds, nthreads=8)
    assert_same(got, ref)


@pytest.mark.parametrize("streams,opts", [(1, {}), (2, {}), (2, {"cu_partition": 4}), (1, {"path_slots": 4096})])
def test_one_draw_per_frame_left_in_flight(pt, oracle, renderer_mod, streams, opts):
    """The reference's call pattern (dispatch.java:693-705: one glDrawArrays per frame, nothing waited for): 100 frames submitted ONE per pt_render_batch_async call.
    The scheduler keeps two groups of iterations in flight and looks at their snapshots when their events have fired, retires up to eight batches per look, makes the
    caller wait for ring rows beyond 64 frames in flight, and lets a second image start underneath — the accumulator must equal the same frames as ONE batch and the oracle's.
    (cu_partition: the CU-masked stream pair of profiles/r06_c_cu_partition.txt, an option that is not the default.)"""
    W, H, N = 64, 36, 100
    wl = pt.scenes.build("C3", W, H)
    seeds = seeds_for(pt, 1, N)
    make = (lambda: renderer_mod.Renderer(W, H, devices=[0] * streams)) if streams > 1 else (lambda: renderer_mod.Renderer(W, H))
    r = make()
    for k, v in opts.items():
        r.set_option(k, v)
    r.load_workload(wl); r.reset_frame()
    for f in range(N):
        r.render_batch_async(f + 1, seeds[f:f + 1])
    r.next_image()                                            # a second image underneath the first one's last paths
    for f in range(3):
        r.render_batch_async(f + 1, seeds[f:f + 1])
    second = r.read_frame().copy()
    first = image_of_age(renderer_mod, r, 1)
    r.close()
    r = make(); r.load_workload(wl); r.reset_frame(); r.render_batch(1, seeds)
    whole = r.read_frame().copy(); r.reset_frame(); r.render_batch(1, seeds[:3]); three = r.read_frame().copy(); r.close()
    assert np.array_equal(first.view(np.uint32), whole.view(np.uint32)) and np.all(first[..., 3] == N)
    assert np.array_equal(second.view(np.uint32), three.view(np.uint32))
    ref, _ = oracle.render_frames(oracle.Scene.from_workload(wl), W, H, 1, 3, seeds[:3], nthreads=8)
    assert_same(second, ref)


def image_of_age(renderer_mod, r, age):
    """the FRAME image `age` pt_next_image calls ago, gathered and copied to the host (hipMemcpy of the ONE runtime this process has mapped)"""
    import ctypes
    ptr = r.gather_image(age)
    r.stream_wait()
    out = np.zeros((r.H, r.W, 4), np.float32)
    hip = ctypes.CDLL(renderer_mod.hip_runtime_info()["path"])
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    assert hip.hipMemcpy(out.ctypes.data, ptr, out.nbytes, 2) == 0
    return out


def test_sharding_invariance(pt, oracle, renderer_mod):
    """K10: 1 vs 2/4 tile shards give bit-identical framebuffers (RNG keyed on the global pixel index)"""
    W, H = 96, 40
    wl = pt.scenes.build("C3", W, H)
    seeds = seeds_for(pt, 1, 2)
    r = renderer_mod.Renderer(W, H)
    r.load_workload(wl); r.reset_frame(); r.render_batch(1, seeds)
    full = r.read_frame().copy(); r.close()
    for count in (2, 4):
        acc = np.zeros_like(full)
        for rank in range(count):
            rr = renderer_mod.Renderer(W, H, shard_rank=rank, shard_count=count)
            rr.load_workload(wl); rr.reset_frame(); rr.render_batch(1, seeds)
            rr.read_frame(acc); rr.close()
        assert np.array_equal(acc, full)


def test_errors_surface(pt, renderer_mod):
    wl = pt.scenes.build("C2", 64, 36)
    r = renderer_mod.Renderer(64, 36)
    r.load_workload(wl)
    bad = wl.buffers[4].copy(); bad[2] = 65.0                 # Parameters.resolution that is not the FRAME image's width
    r.set_buffer(4, bad)
    with pytest.raises(renderer_mod.PtError) as e:
        r.render(1, 1)
    assert e.value.code == -1
    tri = wl.buffers[3].copy(); tri[36] = 99.0
    r.set_buffer(4, wl.buffers[4]); r.set_buffer(3, tri)
    with pytest.raises(renderer_mod.PtError) as e:
        r.render(1, 1)
    assert e.value.code == -4
    # the limit of the slot encoding surfaces as an error, not as a wrong image: SAMPLE_RES beyond the 11-bit counter.  (More distinct refraction indices than
    # the index-stack dictionary holds used to be one too; such a scene now carries the ten floats of the stack: test_more_refraction_indices_than_the_dictionary_holds)
    r.set_buffer(3, wl.buffers[3])
    p = wl.buffers[4].copy(); p[4] = 2048.0
    r.set_buffer(4, p)
    with pytest.raises(renderer_mod.PtError) as e:
        r.render(1, 1)
    assert e.value.code == -1 and "SAMPLE_RES" in str(e.value)
    r.set_buffer(4, wl.buffers[4])
    me = int(wl.buffers[14][0]); n = 300
    mtl = np.zeros(1 + me * n, np.float32); mtl[0] = me
    one = wl.buffers[14][1:1 + me]
    for m in range(n):
        mtl[1 + me * m:1 + me * (m + 1)] = one
        mtl[me * m + 16] = 1.0 + 0.001 * m                   # Ni (dispatch.java:272-324: slot 16 of the record)
    r.set_buffer(14, mtl)
    r.render(1, 1)                                            # no transmissive material: any number of Ni values
    mtl[me * 7 + 12] = 0.5                                    # Tr > 0 on one of them
    r.set_buffer(14, mtl)
    r.render(2, 2)                                            # ... and with one: the float-stack state (parity: the test named above)
    r.close()


def test_render_parity_c4_big_bvh(pt, oracle, renderer_mod):
    """C4: one 100k-triangle BVH (int32 stack entries, deep tree, nodes mostly outside the LDS tile)"""
    wl = pt.scenes.build("C4", 96, 54)
    assert wl.info["triangles"] > 100000
    got, ref, cnt, ocnt = render_both(pt, oracle, renderer_mod, wl, 2)
    assert_same(got, ref, cnt, ocnt)
    got, ref, cnt, ocnt = render_both(pt, oracle, renderer_mod, wl, 1, extend_mode=0)
    assert_same(got, ref, cnt, ocnt)


def test_render_parity_c5_16_bounces(pt, oracle, renderer_mod):
    """C5 materials (clearcoat, subsurface, the reference's "test" material with Tr) at 16 bounces"""
    wl = pt.scenes.build("C5", 96, 54)
    assert wl.max_bounces == 16
    got, ref, cnt, ocnt = render_both(pt, oracle, renderer_mod, wl, 2)
    assert_same(got, ref, cnt, ocnt)


def test_sky_texture_bilinear_repeat(pt, oracle, renderer_mod):
    """K3 on the device: a non-trivial RGBA8 sky through the software LINEAR/REPEAT sampler"""
    wl = pt.scenes.build("C1", 64, 64)
    rs = np.random.RandomState(5)
    wl.sky = rs.randint(0, 256, size=(5, 7, 4)).astype(np.uint8)
    got, ref, cnt, ocnt = render_both(pt, oracle, renderer_mod, wl, 2)
    assert_same(got, ref, cnt, ocnt)


def test_full_size_c3_with_equirect_sky(pt, oracle, renderer_mod):
    """the reference binds an equirect image as texture 0 (dispatch.java:221) and reads it on every miss (frag.glsl:235-242, :877): full C3
    size with a 1024x512 sky and an open room (no ceiling light needed: the sky is the light), on the shipped kernels, against the oracle on
    a pixel lattice"""
    W, H = 1920, 1080
    wl = pt.scenes.build("C3", W, H)
    wl.sky = pt.scenes.equirect_sky(512, 1024)
    b = dict(wl.buffers)
    b[0] = np.array([0.0, 1.0, -2.6], np.float32)        # further back: a third of the primary rays pass the room and see the sky directly
    wl = pt.scenes.Workload(wl.name, W, H, b, wl.sky, wl.sample_res, wl.max_bounces, wl.info)
    seeds = seeds_for(pt, 1, 2)
    r = renderer_mod.Renderer(W, H, devices=[0, 0])      # the production form: two streams on the GPU
    r.load_workload(wl); r.reset_frame()
    r.render_batch(1, seeds)
    a = r.read_frame().copy()
    r.close()
    sc = oracle.Scene.from_workload(wl)
    ref = np.zeros((H, W, 4), np.float32)
    for i, sd in enumerate(seeds):
        oracle.render(sc, W, H, 1 + i, sd, ref, nthreads=8, xs=24, ys=27)
    assert np.array_equal(a[::27, ::24], ref[::27, ::24])
    lattice = a[::27, ::24, :3] / 2.0
    assert len(np.unique(np.round(lattice.reshape(-1, 3), 4), axis=0)) > 1000      # the sky really varies over the image
    # and the small-size form through the counting kernels too
    bs = dict(b); bs[4] = pt.scenes.make_params(96, 54, wl.sample_res, wl.max_bounces)
    small = pt.scenes.Workload(wl.name, 96, 54, bs, wl.sky, wl.sample_res, wl.max_bounces, wl.info)
    got, ref, cnt, ocnt = render_both(pt, oracle, renderer_mod, small, 2)
    assert_same(got, ref, cnt, ocnt)


NO_VN_CUBE = """v -1 -1 -1\nv 1 -1 -1\nv 1 1 -1\nv -1 1 -1\nv -1 -1 1\nv 1 -1 1\nv 1 1 1\nv -1 1 1
f 1 2 3\nf 1 3 4\nf 5 7 6\nf 5 8 7\nf 1 5 6\nf 1 6 2\nf 4 3 7\nf 4 7 8\nf 1 4 8\nf 1 8 5\nf 2 6 7\nf 2 7 3\n"""


def _no_vn_workload(pt, W, H):
    """the shape of the reference's only shipped asset (src/objs/table - Copy.obj: five `o` cubes, `v` and `f a b c` lines only — no vn, vt or
    usemtl) plus two scene.addTri triangles (dispatch.java:1013-1015): every such triangle gets normalize(vec(0)) = NaN normals
    (dispatch.java:901-902, 1241-1243), uploaded as they are.  (addTri only appends to the triangle list: such a triangle is
    in no BVH and no ray ever meets it, in the reference as here; the cubes are what the paths hit.)  Text generated here, not the reference's file."""
    S = pt.scenes
    sc = S._new_scene()
    sc.addMaterial("default"); sc.setLastMtl("Kd", (0.8, 0.8, 0.8)); sc.setLastMtl("Pr", 1)
    sc.addMaterial("glassy"); sc.setLastMtl("Tr", 0.8); sc.setLastMtl("Ni", 1.4); sc.setLastMtl("Pr", 1)
    lines, base = [], 0
    for k, (cx, cy, cz, h) in enumerate([(0.0, 0.5, 3.0, 0.5), (1.4, 0.3, 2.6, 0.3), (-1.3, 0.35, 2.4, 0.35), (0.6, 0.2, 1.7, 0.2), (-0.5, 1.4, 3.4, 0.25)]):
        lines.append(f"o cube{k}")
        for ln in NO_VN_CUBE.replace("\\n", "\n").splitlines():
            t = ln.split()
            if t[0] == "v":
                lines.append("v %.9g %.9g %.9g" % (cx + h * float(t[1]), cy + h * float(t[2]), cz + h * float(t[3])))
            else:
                lines.append("f %d %d %d" % tuple(int(q) + base for q in t[1:]))
        base += 8
    sc.addObjectText("\n".join(lines) + "\n", 0, parentDirectory="")
    # a ground quad WITH normals, so that finite and NaN pixels sit side by side
    o = S.Obj(); o.group("ground"); o.quad((-6, 0, -2), (6, 0, -2), (6, 0, 10), (-6, 0, 10), (0, 1, 0))
    sc.addObjectText(o.text(), 0, parentDirectory="")
    sc.addTri((-2.5, 0.1, 4.0), (-1.5, 0.1, 4.0), (-2.0, 1.6, 4.2), 0)
    sc.addTri((1.8, 0.1, 3.8), (2.8, 0.1, 3.8), (2.3, 1.2, 3.6), 1)
    return S._finish("no_vn", sc, W, H, (0.0, 0.9, -0.5), (0.08, 0.0, 0.0), (153, 179, 230), 4, 6)


def test_no_vn_geometry_renders_nan_like_the_oracle(pt, oracle, renderer_mod):
    """Q-5: OBJ groups without `vn` and scene.addTri triangles carry NaN normals; a path that hits one continues along a NaN direction
    (misses everything, samples the sky at NaN coordinates) and its pixel is NaN in the reference.  NaN == NaN against the oracle, on the
    counting and on the shipped kernels, one stream and two."""
    wl = _no_vn_workload(pt, 96, 54)
    tri = wl.buffers[3].reshape(-1, 40)
    assert np.isnan(tri[:60, 12:15]).all() and np.isnan(tri[-2:, 12:15]).all() and np.isfinite(tri[60:62, 12:15]).all()
    got, ref, cnt, ocnt = render_both(pt, oracle, renderer_mod, wl, 3)
    assert_same(got, ref, cnt, ocnt)
    nan_px = np.isnan(ref[..., :3]).any(axis=2)
    assert 0.05 < nan_px.mean() < 0.9, nan_px.mean()          # cubes and triangles are NaN, ground and sky are not
    r = renderer_mod.Renderer(wl.W, wl.H, devices=[0, 0])
    r.load_workload(wl); r.reset_frame(); r.render_batch(1, seeds_for(pt, 1, 3))
    two = r.read_frame(); r.close()
    assert_same(two, ref)


def test_rotated_camera_and_rotated_ellipsoid(pt, oracle, renderer_mod):
    """rotate()/rotateBack() paths (frag.glsl:244-297, Q-15) and a camera with all three angles set"""
    wl = pt.scenes.build("C1", 64, 48)
    b = dict(wl.buffers)
    b[1] = np.array([0.2, -0.3, 0.1], np.float32)
    e = b[7].copy(); n = int(e[0]); e[1 + 6 * n: 1 + 6 * n + 3] = [0.3, 0.2, 0.1]     # rotate the first ellipsoid
    b[7] = e
    wl2 = pt.scenes.Workload(wl.name, wl.W, wl.H, b, wl.sky, wl.sample_res, wl.max_bounces, wl.info)
    got, ref, cnt, ocnt = render_both(pt, oracle, renderer_mod, wl2, 2)
    assert_same(got, ref, cnt, ocnt)


def test_full_size_properties_c3(pt, oracle, renderer_mod):
    """BASELINE.json's full C3 size (1920x1080, 8 spp/frame): size-independent properties + oracle parity on a pixel lattice"""
    W, H = 1920, 1080
    wl = pt.scenes.build("C3", W, H)
    seeds = seeds_for(pt, 1, 4)
    r = renderer_mod.Renderer(W, H)
    r.load_workload(wl); r.reset_frame()
    r.render_batch(1, seeds)
    a = r.read_frame().copy()
    r.reset_frame()
    r.render_batch(1, seeds[:1]); r.render_batch(2, seeds[1:3]); r.render(4, seeds[3])     # any split of the batch: same bits
    b = r.read_frame().copy()
    r.close()
    assert np.array_equal(a, b)
    assert np.all(a[..., 3] == 4.0) and np.isfinite(a[..., :3]).all() and (a[..., :3] >= 0).all()
    # the oracle on every 24th pixel in x and 27th in y, same frames
    sc = oracle.Scene.from_workload(wl)
    ref = np.zeros((H, W, 4), np.float32)
    for i, s in enumerate(seeds):
        oracle.render(sc, W, H, 1 + i, s, ref, nthreads=8, xs=24, ys=27)
    assert np.array_equal(a[::27, ::24], ref[::27, ::24])


def test_full_size_c1_whole_image(pt, oracle, renderer_mod):
    """BASELINE.json configs[0] (C1: 256x256, SAMPLE_RES 4, 1 frame, 4 bounces, ellipsoids + ground quad) in full: every pixel against the oracle"""
    cfg = pt.scenes.CONFIGS["C1"]
    wl = pt.scenes.build("C1", cfg["W"], cfg["H"])
    assert (wl.sample_res, wl.max_bounces) == (4, 4)
    got, ref, cnt, ocnt = render_both(pt, oracle, renderer_mod, wl, cfg["spp"] // cfg["sample_res"])
    assert_same(got, ref, cnt, ocnt)


def test_full_size_c2_lattice_and_display(pt, oracle, renderer_mod):
    """BASELINE.json configs[1] (C2: 1280x720, 64 spp = 8 frames x 8, Cornell box) in full: oracle on a pixel lattice, and the 8-bit display
    image of the whole frame derived from the same accumulator"""
    cfg = pt.scenes.CONFIGS["C2"]
    W, H = cfg["W"], cfg["H"]
    wl = pt.scenes.build("C2", W, H)
    n = cfg["spp"] // cfg["sample_res"]
    seeds = seeds_for(pt, 1, n)
    r = renderer_mod.Renderer(W, H)
    r.load_workload(wl); r.reset_frame(); r.render_batch(1, seeds)
    a = r.read_frame().copy()
    disp = r.read_display(n, java_bytes=False)
    r.close()
    sc = oracle.Scene.from_workload(wl)
    ref = np.zeros((H, W, 4), np.float32)
    for i, sd in enumerate(seeds):
        oracle.render(sc, W, H, 1 + i, sd, ref, nthreads=8, xs=16, ys=18)
    assert np.array_equal(a[::18, ::16], ref[::18, ::16])
    assert np.all(a[..., 3] == n)
    assert np.array_equal(disp, oracle.display(a, n, java_bytes=False))


@pytest.mark.parametrize("name,xs,ys", [("C4", 24, 27), ("C5", 48, 54), ("C6", 48, 54)])
def test_full_size_properties_c4_c5(pt, oracle, renderer_mod, name, xs, ys):
    """BASELINE.json's full C4 (1920x1080, one 100k-triangle BVH) and C5 (3840x2160, 16 bounces) sizes, and C6 (1920x1080, 64 BVHs + 3 ellipsoids): the overlapped schedule and a
    3-way tile split reproduce the synchronous image bit for bit; oracle parity on a pixel lattice"""
    cfg = pt.scenes.CONFIGS[name]
    W, H = cfg["W"], cfg["H"]
    wl = pt.scenes.build(name, W, H)
    seeds = seeds_for(pt, 1, 2)
    r = renderer_mod.Renderer(W, H)
    r.load_workload(wl); r.reset_frame()
    r.render_batch(1, seeds)
    a = r.read_frame().copy()
    r.next_image()
    r.render_batch_async(1, seeds[:1]); r.render_batch_async(2, seeds[1:])
    b = r.read_frame().copy()
    r.close()
    assert np.array_equal(a, b)
    assert np.all(a[..., 3] == 2.0)
    acc = np.zeros_like(a)
    for rank in range(3):
        rr = renderer_mod.Renderer(W, H, shard_rank=rank, shard_count=3)
        rr.load_workload(wl); rr.reset_frame(); rr.render_batch(1, seeds)
        part = rr.read_frame(); rr.close()
        assert np.all(acc[part[..., 3] > 0] == 0)           # shards are disjoint
        acc += part
    assert np.array_equal(acc, a)
    sc = oracle.Scene.from_workload(wl)
    ref = np.zeros((H, W, 4), np.float32)
    for i, sd in enumerate(seeds):
        oracle.render(sc, W, H, 1 + i, sd, ref, nthreads=8, xs=xs, ys=ys)
    assert np.array_equal(a[::ys, ::xs], ref[::ys, ::xs])


@pytest.mark.parametrize("slots,batch", [(2048, 3), (1 << 16, 1), (4096, 6), (0, 1), (0, 2)])
def test_overlapped_batches_equal_synchronous(pt, oracle, renderer_mod, slots, batch):
    """pt_render_batch_async / pt_next_image / pt_finish_image: consecutive batches share one running path pool (no drain between
    them; with (65536, 1) the pool is larger than a batch and runs dry in between, so dead slots are revived) - same bits as the
    synchronous calls, image after image.  slots 0 = automatic: the pool starts at the size of the first small batch and grows with
    the backlog of the following submissions"""
    import torch
    from pathtracer_0_amd import shard
    W, H = 128, 72
    wl = pt.scenes.build("C3", W, H)
    sc = oracle.Scene.from_workload(wl)
    dev = torch.device("cuda", 0)
    r = renderer_mod.Renderer(W, H)
    if slots:
        r.set_option("path_slots", slots)
    r.set_option("count_stats", 1)
    r.load_workload(wl)
    images = []
    n_img, n_frames = 3, 6
    seeds = {k: [(977 * k + 31 * f) % 10000 for f in range(1, n_frames + 1)] for k in range(n_img)}
    for k in range(n_img):
        r.next_image()
        for first in range(1, n_frames + 1, batch):
            r.render_batch_async(first, seeds[k][first - 1:first - 1 + batch])
        if k > 0:
            r.finish_image(1)
            images.append(shard.frame_tensor(r, dev, age=1).cpu().numpy().reshape(H, W, 4).copy())
    images.append(r.read_frame().copy())          # completes the last image
    cnt = r.counters()
    r.close()
    total = 0
    for k in range(n_img):
        ref, ocnt = oracle.render_frames(sc, W, H, 1, n_frames, seeds[k], nthreads=8)
        assert np.array_equal(images[k], ref), k
        total += int(ocnt[4])
    assert cnt["samples"] == total


def test_overlapped_batches_with_a_camera_move(pt, oracle, renderer_mod):
    """a change of the frame inputs between two asynchronous batches ends the running stream first (include/pt_api.h)"""
    W, H = 96, 54
    wl = pt.scenes.build("C2", W, H)
    r = renderer_mod.Renderer(W, H)
    r.set_option("path_slots", 1024)
    r.load_workload(wl)
    r.reset_frame()
    r.render_batch_async(1, [11, 22])
    moved = wl.buffers[0].copy(); moved[0] += 0.25
    r.set_buffer(0, moved)
    r.render_batch_async(3, [33, 44])
    got = r.read_frame().copy()
    r.close()
    sc = oracle.Scene.from_workload(wl)
    ref = np.zeros((H, W, 4), np.float32)
    oracle.render_frames(sc, W, H, 1, 2, [11, 22], frame=ref, nthreads=8)
    b = dict(wl.buffers); b[0] = moved
    sc2 = oracle.Scene.from_workload(pt.scenes.Workload(wl.name, W, H, b, wl.sky, wl.sample_res, wl.max_bounces, wl.info))
    oracle.render_frames(sc2, W, H, 3, 2, [33, 44], frame=ref, nthreads=8)
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("name,change", [("C5", dict(RAYTRACING=0)), ("C3", dict(SAMPLE_RES=4, MAX_BOUNCES=2)), ("C3", dict(SAMPLE_RES=8, MAX_BOUNCES=12)), ("C5", dict(RAYTRACING=0, SAMPLE_RES=2))])
def test_parameter_upload_between_overlapped_batches(pt, oracle, renderer_mod, name, change):
    """pt_set_buffer(PT_BIND_PARAMS) while asynchronous batches are in flight (the reference's own loop drops SAMPLE_RES / MAX_BOUNCES to 4 / 2
    while the camera moves, dispatch.java:646-666): the running stream finishes with the Parameters it was started with — kernel variant
    (RAYTRACING), loop bounds, iteration budget — and the next batch starts a new stream with the new block"""
    W, H = 96, 54
    wl = pt.scenes.build(name, W, H)
    wl2 = wl.with_params(**change)
    r = renderer_mod.Renderer(W, H)
    r.set_option("path_slots", 2048)                           # small pool: a deep backlog is still unissued when the upload arrives
    r.load_workload(wl)
    r.reset_frame()
    r.render_batch_async(1, [11, 22, 33])
    r.set_buffer(4, wl2.buffers[4])                            # no flush here: the stream keeps running on the old block
    r.render_batch_async(4, [44, 55])
    r.set_buffer(4, wl.buffers[4])
    r.render_batch_async(6, [66])
    got = r.read_frame().copy()
    r.close()
    ref = np.zeros((H, W, 4), np.float32)
    oracle.render_frames(oracle.Scene.from_workload(wl), W, H, 1, 3, [11, 22, 33], frame=ref, nthreads=8)
    oracle.render_frames(oracle.Scene.from_workload(wl2), W, H, 4, 2, [44, 55], frame=ref, nthreads=8)
    oracle.render_frames(oracle.Scene.from_workload(wl), W, H, 6, 1, [66], frame=ref, nthreads=8)
    assert_same(got, ref)


def test_debug_probe_does_not_disturb_batches_in_flight(pt, oracle, renderer_mod):
    """pt_debug_intersect rewrites the frame constants (it forces AUTO_FOCUS = 0): work in flight is finished first"""
    W, H = 96, 54
    wl = pt.scenes.build("C3", W, H)
    r = renderer_mod.Renderer(W, H)
    r.set_option("path_slots", 1024)
    r.load_workload(wl); r.reset_frame()
    r.render_batch_async(1, [5, 6, 7])
    r.debug_intersect(np.array([[0.0, 1.0, -3.0]], np.float32), np.array([[0.0, 0.0, 1.0]], np.float32))
    r.render_batch_async(4, [8])
    got = r.read_frame().copy()
    r.close()
    ref, _ = oracle.render_frames(oracle.Scene.from_workload(wl), W, H, 1, 4, [5, 6, 7, 8], nthreads=8)
    assert_same(got, ref)


def test_two_process_shards_on_one_gpu(pt):
    """one process per shard (as on the 8-GPU node), here 2 ranks sharing cuda:0, gloo collective"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29533",
                          os.path.join(root, "tests", "_dist_gpu_worker.py")], capture_output=True, text=True, timeout=600, env=env)
    assert "DIST_GPU_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_big_leaves_and_empty_leaf(pt, oracle, renderer_mod):
    """hand-made BVH buffers: a leaf with many triangles (the reference's builder makes those, Q-11), and a node
    with one null child, which the shader treats as a leaf without triangles (bitwise OR test, frag.glsl:478)"""
    W, H = 64, 48
    wl = pt.scenes.build("C2", W, H)
    b = dict(wl.buffers)
    tris = b[3].reshape(-1, 40)
    n = len(tris)
    lo = tris[:, [0, 1, 2, 4, 5, 6, 8, 9, 10]].reshape(-1, 3).min(0); hi = tris[:, [0, 1, 2, 4, 5, 6, 8, 9, 10]].reshape(-1, 3).max(0)
    half = n // 2
    def box(idx):
        p = tris[idx][:, [0, 1, 2, 4, 5, 6, 8, 9, 10]].reshape(-1, 3)
        return list(p.min(0)) + list(p.max(0))
    # node 0: root (children 1, 2); node 1: leaf with `half` triangles; node 2: inner (children 3, -1 -> "leaf" without triangles per :478);
    # node 3 would hold the rest but is unreachable through node 2 -> also list it under a second object root 4 (leaf)
    data = np.zeros((5, 8), np.float32)
    data[0, :6] = list(lo) + list(hi)
    data[1, :6] = box(np.arange(half)); data[1, 6:] = [0, half]
    data[2, :6] = box(np.arange(half, n))
    data[3, :6] = box(np.arange(half, n)); data[3, 6:] = [half, n]
    data[4, :6] = box(np.arange(half, n)); data[4, 6:] = [half, n]
    tree = np.array([[0, 1, 2], [1, -1, -1], [2, 3, -1], [3, -1, -1], [4, -1, -1]], np.int32)
    b[10] = data.reshape(-1); b[11] = tree.reshape(-1); b[12] = np.arange(n, dtype=np.int32); b[13] = np.array([2, 0, 4], np.int32)
    wl2 = pt.scenes.Workload(wl.name, W, H, b, wl.sky, wl.sample_res, wl.max_bounces, wl.info)
    for mode in (0, 1):
        got, ref, cnt, ocnt = render_both(pt, oracle, renderer_mod, wl2, 2, extend_mode=mode)
        assert_same(got, ref, cnt, ocnt)


@pytest.mark.parametrize("kw", [dict(SAMPLE_RES=1, MAX_BOUNCES=1), dict(SAMPLE_RES=3, MAX_BOUNCES=2.5), dict(AUTO_FOCUS=0, FOCAL_DISTANCE=2.5, BLUR=0.05),
                                dict(SAMPLE_RES=300, MAX_BOUNCES=2), dict(SAMPLE_RES=2, MAX_BOUNCES=700.5),
                                dict(BLUR=0.0), dict(screenSize=0.7, focalLength=1.3)])
def test_parameter_block_variants(pt, oracle, renderer_mod, kw):
    """Parameters block (frag.glsl:39-52): float loop bounds (:820,:898), auto-focus off, lens blur, field of view"""
    wl = pt.scenes.build("C3", 100, 37).with_params(**kw)      # 100x37: not a multiple of the 32x8 tile
    got, ref, cnt, ocnt = render_both(pt, oracle, renderer_mod, wl, 2)
    assert_same(got, ref, cnt, ocnt)


def test_odd_size_sharded(pt, oracle, renderer_mod):
    W, H = 100, 37
    wl = pt.scenes.build("C2", W, H)
    seeds = seeds_for(pt, 1, 2)
    ref, _ = oracle.render_frames(oracle.Scene.from_workload(wl), W, H, 1, 2, seeds, nthreads=8)
    acc = np.zeros_like(ref)
    for rank in range(3):
        rr = renderer_mod.Renderer(W, H, shard_rank=rank, shard_count=3)
        rr.load_workload(wl); rr.reset_frame(); rr.render_batch(1, seeds)
        rr.read_frame(acc); rr.close()
    assert np.array_equal(acc, ref)


def test_empty_scene_and_mouse_overlay(pt, oracle, renderer_mod):
    """K2 on the device (no objects: every pixel = sky) and the mouse-probe region (frag.glsl:888-893) leaving FRAME untouched"""
    W, H = 64, 48
    sc = pt.hostlib.Scene(); sc.addMaterial("default")
    b = sc.pack()
    b[0] = np.zeros(3, np.float32); b[1] = np.array([0.1, 0.2, 0.0], np.float32); b[2] = np.array([20.0, 30.0, 0.0], np.float32)
    b[4] = pt.scenes.make_params(W, H, 2, 4)
    sky = np.array([[[10, 200, 90, 255], [250, 3, 77, 255]]], np.uint8)
    wl = pt.scenes.Workload("empty", W, H, b, sky, 2, 4, {})
    got, ref, cnt, ocnt = render_both(pt, oracle, renderer_mod, wl, 2)
    assert_same(got, ref)        # (the device also traces the overlay pixels and drops them at accumulation: counters differ by those)
    assert np.all(got[30, 20] == 0) and np.all(got[0, 0, :3] > 0) and cnt["segments"] == cnt["samples"]


def test_n4_screenshot_png(pt, renderer_mod, tmp_path):
    """functions.screenshot's file (dispatch.java:840-848): the library's PNG decodes (PIL) to exactly the display image, with and without the
    Java signed-byte packing, also from a two-stream context"""
    from PIL import Image
    wl = pt.scenes.build("C3", 100, 37)
    for kw in (dict(), dict(devices=[0, 0])):
        r = renderer_mod.Renderer(100, 37, **kw)
        r.load_workload(wl); r.reset_frame(); r.render_batch(1, seeds_for(pt, 1, 2))
        for jb in (True, False):
            path = tmp_path / f"shot_{jb}.png"
            r.screenshot(path, 2, java_bytes=jb)
            img = np.array(Image.open(path))
            assert img.shape == (37, 100, 3) and img.dtype == np.uint8
            assert np.array_equal(img, r.read_display(2, java_bytes=jb))
        r.close()


def test_n4_display_path(pt, oracle, renderer_mod):
    """8-bit screenshot path on the device == the oracle's, with and without the Java signed-byte packing"""
    W, H = 96, 54
    wl = pt.scenes.build("C3", W, H)
    seeds = seeds_for(pt, 1, 3)
    r = renderer_mod.Renderer(W, H)
    r.load_workload(wl); r.reset_frame(); r.render_batch(1, seeds)
    frame = r.read_frame()
    for jb in (False, True):
        got = r.read_display(3, java_bytes=jb)
        assert np.array_equal(got, oracle.display(frame, 3, java_bytes=jb))
    r.close()


@pytest.mark.parametrize("name,mode", [("C3", 0), ("C3", 1), ("C5", 0), ("C5", 1), ("C1", 0), ("C1", 1), ("C3", 2), ("C6", 2), ("C5", 2), ("C1", 2)])
def test_n2_direct_diffuse_mode(pt, oracle, renderer_mod, name, mode):
    """RAYTRACING == 0 (directDiffuse, frag.glsl:655-681): one segment per sample, the thickness probe for
    subsurface materials (C5), and an ellipsoid with a subsurface material (parentID = -1: probe treated as a miss)"""
    kw = dict(subdiv=2) if name == "C5" else dict(nu=8, nv=8) if name == "C6" else {}
    wl = pt.scenes.build(name, 96, 54, **kw).with_params(RAYTRACING=0)
    if name == "C1":
        b = dict(wl.buffers); m = b[14].copy()
        m[1 + 48 + 42 - 1 + 0] = 0.5                      # material 1: subsurface
        m[1 + 48 + 43 - 1: 1 + 48 + 46 - 1] = [0.4, 0.8, 0.5]; m[1 + 48 + 46 - 1: 1 + 48 + 49 - 1] = [1, 1, 1]
        b[14] = m
        wl = pt.scenes.Workload(wl.name, wl.W, wl.H, b, wl.sky, wl.sample_res, wl.max_bounces, wl.info)
    # (round 5: directDiffuse runs on the hand-written intersect kernel, thickness probes included: FL_PROBE rays are set up at its refill)
    got, ref, cnt, ocnt = render_both(pt, oracle, renderer_mod, wl, 2, extend_mode=mode, expect_asm=(mode == 2))
    assert_same(got, ref, cnt, ocnt)
    assert cnt["segments"] == cnt["samples"]


@pytest.mark.parametrize("raytracing", [1, 0])
def test_n3_texture_maps(pt, oracle, renderer_mod, raytracing):
    """material texture maps (mapMtl, frag.glsl:210-225) + the raw-texel normal map (:827), in both render modes"""
    wl = pt.scenes.build("T1", 96, 54).with_params(RAYTRACING=raytracing)
    got, ref, cnt, ocnt = render_both(pt, oracle, renderer_mod, wl, 3)
    assert_same(got, ref, cnt, ocnt)
    # the maps matter: the same scene without its textures renders differently
    b = dict(wl.buffers); m = b[14].copy().reshape(-1)
    for mat in range((len(m) - 1) // 48):
        for k in (22, 23, 24, 32, 33, 34, 35, 36, 37, 38, 39, 40, 41):
            m[48 * mat + k] = -1.0
    b[14] = m
    plain = pt.scenes.Workload(wl.name, wl.W, wl.H, b, wl.sky, wl.sample_res, wl.max_bounces, wl.info)
    got2, ref2, _, _ = render_both(pt, oracle, renderer_mod, plain, 3, count_stats=False)
    assert_same(got2, ref2)
    assert rmse(got, got2) > 1e-2


def test_n3_asset_directory(pt, oracle, renderer_mod, tmp_path):
    """the reference's loading path end to end: addObject(<directory>) -> parseMtls -> texture table -> render (dispatch.java:869-882)"""
    from test_host_scene import _asset_dir
    wl = pt.scenes.asset_workload(_asset_dir(tmp_path), 80, 48, sample_res=4, max_bounces=6)
    assert sorted(wl.textures) == [1, 2, 3]
    got, ref, cnt, ocnt = render_both(pt, oracle, renderer_mod, wl, 2)
    assert_same(got, ref, cnt, ocnt)
    assert got[..., :3].max() > 0


# ---- the BVH builder on the GPU (pt_build_bvh) against the CPU mirror of the Java builder
@pytest.mark.parametrize("name", ["C2", "C3", "C4", "C5"])
def test_gpu_bvh_builder_equals_cpu_builder(pt, name):
    cpu = pt.scenes.build(name, 64, 36)
    pt.scenes.GPU_BVH = 0
    try:
        gpu = pt.scenes.build(name, 64, 36)
    finally:
        pt.scenes.GPU_BVH = None
    assert gpu.info == cpu.info
    for b in (3, 10, 11, 12, 13):
        assert np.array_equal(cpu.buffers[b].view(np.uint32), gpu.buffers[b].view(np.uint32)), b


def _soup_obj(rs, n, degenerate=0.0):
    """n random small triangles in the unit cube; a fraction share one centroid (leaves with several triangles)"""
    lines = ["o soup", "vn 0 0 1"]
    k = 0
    for i in range(n):
        c = rs.rand(3) if rs.rand() >= degenerate else np.array([0.5, 0.5, 0.5])
        d = (rs.rand(3, 3) - 0.5) * 0.05
        d -= d.mean(axis=0)                                        # exact-ish common centroid for the degenerate ones
        for v in c + d:
            lines.append("v %.17g %.17g %.17g" % tuple(v))
        lines.append("f %d//1 %d//1 %d//1" % (k + 1, k + 2, k + 3))
        k += 3
    return "\n".join(lines) + "\n"


@pytest.mark.parametrize("n,degenerate,seed", [(2, 0.0, 1), (3, 0.0, 2), (65, 0.0, 3), (64, 0.5, 4), (1000, 0.0, 5), (5000, 0.3, 6), (40000, 0.02, 7)])
def test_gpu_bvh_builder_random_soups(pt, n, degenerate, seed):
    text = _soup_obj(np.random.RandomState(seed), n, degenerate)
    out = []
    for gpu in (False, True):
        sc = pt.hostlib.Scene(); sc.addMaterial("m")
        if gpu:
            sc.use_gpu_bvh_builder(0)
        sc.addObjectText(text, 0)
        sc.addObjectText(text, 0, shift=(2.0, 0.0, 0.0))          # a second object: ids and leaf offsets continue
        out.append((sc.pack(), {k: sc.count(k) for k in ("nodes", "objects", "max_depth", "max_leaf", "leaf_indices")}))
    assert out[0][1] == out[1][1]
    for b in (3, 10, 11, 12, 13):
        assert np.array_equal(out[0][0][b].view(np.uint32), out[1][0][b].view(np.uint32)), b


def test_gpu_bvh_builder_errors(pt):
    sc = pt.hostlib.Scene(); sc.addMaterial("m"); sc.use_gpu_bvh_builder(0)
    with pytest.raises(RuntimeError, match="Q-16"):
        sc.addObjectText("o one\nv 0 0 0\nv 1 0 0\nv 0 1 0\nvn 0 0 1\nf 1//1 2//1 3//1\n", 0)


def test_n3_missing_texture_is_an_error(pt, renderer_mod):
    wl = pt.scenes.build("T1", 64, 36)
    r = renderer_mod.Renderer(64, 36)
    for bnd, arr in wl.buffers.items():
        r.set_buffer(bnd, arr)
    r.set_texture(0, wl.sky)                      # material textures 1..4 never uploaded
    with pytest.raises(renderer_mod.PtError) as e:
        r.render(1, 1)
    assert e.value.code == -4
    r.close()


@pytest.mark.parametrize("raytracing,mode", [(1, 2), (1, 1), (1, 0), (0, 1)])
def test_n3_mapped_material_on_ellipsoids(pt, oracle, renderer_mod, raytracing, mode):
    """a texture-mapped material on an ellipsoid is sampled at the uv of the closest TRIANGLE the BVH loop found before it (hitUV is
    written at frag.glsl:574 only, not at :619-630), (0,0) when no triangle lies on the ray: C1's spheres over its ground quad"""
    wl1 = pt.scenes.build("C1", 96, 96)
    b = dict(wl1.buffers); m = b[14].copy()
    m[23] = 1.0                                   # material 0 (ground AND an ellipsoid): map_Kd = texture 1
    m[48 + 33] = 2.0                              # material 1 (ellipsoids only): map_Pr = texture 2
    b[14] = m
    rs = np.random.RandomState(4)
    tex = {1: rs.randint(0, 256, size=(5, 7, 4)).astype(np.uint8), 2: rs.randint(0, 256, size=(3, 2, 4)).astype(np.uint8)}
    wl = pt.scenes.Workload(wl1.name + "_mapped", wl1.W, wl1.H, b, wl1.sky, wl1.sample_res, wl1.max_bounces, dict(wl1.info), tex)
    wl = wl.with_params(RAYTRACING=raytracing)
    # (round 5: path tracing such a scene runs on the hand-written intersect kernel too — it leaves the triangle's (u, v, id) in the side record State::HX)
    got, ref, cnt, ocnt = render_both(pt, oracle, renderer_mod, wl, 3, extend_mode=mode, expect_asm=(mode == 2 and raytracing == 1))
    assert_same(got, ref, cnt, ocnt)
    plain, _, _, _ = render_both(pt, oracle, renderer_mod, wl1.with_params(RAYTRACING=raytracing), 3, extend_mode=mode)
    assert not np.array_equal(got, plain)         # the maps do change the picture


# ---- randomized scenes: every material lobe, texture maps, ellipsoids (stretched, rotated), several objects, odd cameras — in combination
def _random_workload(pt, seed, W=72, H=44, ellipsoid_maps=False, many_groups=False):
    rs = np.random.RandomState(seed)
    sc = pt.hostlib.Scene()
    names = []
    n_mat = rs.randint(3, 7)
    textures = {}
    for m in range(n_mat):
        name = f"m{m}"; names.append(name)
        sc.addMaterial(name)
        sc.setLastMtl("Kd", rs.uniform(0.2, 0.95, 3)); sc.setLastMtl("Ks", rs.uniform(0.2, 1.0, 3))
        sc.setLastMtl("Pr", float(rs.choice([1.0, rs.uniform(0.0, 1.0)]))); sc.setLastMtl("Pm", float(rs.choice([0.0, 1.0, rs.uniform(0, 1)])))
        sc.setLastMtl("Pc", float(rs.choice([0.0, rs.uniform(0, 0.8)]))); sc.setLastMtl("Pcr", float(rs.uniform(0, 0.5)))
        kind = rs.randint(0, 6)
        if kind == 0:
            sc.setLastMtl("Ke", rs.uniform(1.0, 12.0, 3))
        elif kind == 1:
            sc.setLastMtl("Tr", float(rs.uniform(0.3, 1.0))); sc.setLastMtl("Ni", float(rs.uniform(1.05, 1.9))); sc.setLastMtl("Tf", rs.uniform(0.0, 0.6, 3))
            sc.setLastMtl("Density", float(rs.uniform(0.2, 3.0)))
        elif kind == 2:
            sc.setLastMtl("Tf", rs.uniform(0.1, 0.9, 3)); sc.setLastMtl("Ni", float(rs.uniform(1.1, 1.6)))
        elif kind == 3:
            sc.setLastMtl("illum", int(rs.choice([5, 7]))); sc.setLastMtl("Ni", float(rs.uniform(1.1, 2.4)))
        elif kind == 4:
            sc.setLastMtl("subsurface", float(rs.uniform(0.2, 0.9))); sc.setLastMtl("subsurfaceColor", rs.uniform(0.1, 1, 3)); sc.setLastMtl("subsurfaceRadius", rs.uniform(0.1, 1, 3))
            sc.setLastMtl("Ka", rs.uniform(0, 0.2, 3))
        if (m >= 2 or ellipsoid_maps) and rs.rand() < 0.5:          # maps on triangle-only materials; with ellipsoid_maps also on the ellipsoids' 0 and 1
            for field in rs.choice(["map_Kd", "map_Ks", "map_Ke", "map_Pr", "map_Pm", "map_Pc", "map_Tr", "map_Ka", "map_bump"], size=rs.randint(1, 4), replace=False):
                t = len(textures) + 1
                textures[t] = rs.randint(0, 256, size=(rs.randint(1, 9), rs.randint(1, 9), 4)).astype(np.uint8)
                sc.setLastMtl(str(field), t)
    o = pt.scenes.Obj()
    o.group("ground"); o.usemtl(names[0])
    o.quad_uv((-3, -0.5, -3), (3, -0.5, -3), (3, -0.5, 3), (-3, -0.5, 3), (0, 1, 0), (0.0, 0.0), (2.5, 2.5))
    # many_groups: 9 .. 70 small `o` groups (more than 8 BVHs: root records in LDS and the per-ray cull of the object loop; beyond 64: two BVHs per mask bit)
    for g in range(rs.randint(9, 71) if many_groups else rs.randint(1, 4)):
        o.group(f"soup{g}")
        centre = rs.uniform(-1.0, 1.0, 3) + np.array([0, 0.4, 0.5])
        for _ in range(rs.randint(8, 30) if many_groups else rs.randint(8, 120)):
            o.usemtl(names[rs.randint(0, n_mat)])
            c = centre + rs.normal(0, 0.45, 3)
            a, b2, c2 = (c + rs.normal(0, 0.18, 3) for _ in range(3))
            nrm = np.cross(b2 - a, c2 - a); nrm = nrm / (np.linalg.norm(nrm) + 1e-12)
            if rs.rand() < 0.5:
                nrm = nrm + rs.normal(0, 0.2, 3)                    # a "smooth" normal that is not the face normal; components are non-zero -> interpolated branch
            i0 = o.v(a); i1 = o.v(b2); i2 = o.v(c2); k = o.vn(nrm)
            if rs.rand() < 0.7:
                t0 = o.vt(rs.uniform(0.05, 3.0, 2)); t1 = o.vt(rs.uniform(0.05, 3.0, 2)); t2 = o.vt(rs.uniform(0.05, 3.0, 2))
                o.lines.append(f"f {i0}/{t0}/{k} {i1}/{t1}/{k} {i2}/{t2}/{k}")
            else:
                o.lines.append(f"f {i0}//{k} {i1}//{k} {i2}//{k}")
    sc.addObjectText(o.text(), 0, parentDirectory="")
    for e in range(rs.randint(0, 4)):
        rot = rs.uniform(-0.6, 0.6, 3) if rs.rand() < 0.5 else (0.0, 0.0, 0.0)
        sc.addEllipsoid(rs.uniform(-1.2, 1.2, 3) + np.array([0, 0.3, 0.8]), rs.uniform(0.5, 2.0, 3), rot, float(rs.uniform(0.15, 0.5)), int(rs.randint(0, 2)))
    cam = (float(rs.uniform(-0.5, 0.5)), float(rs.uniform(0.2, 1.2)), float(rs.uniform(-3.0, -1.8)))
    rot = (float(rs.uniform(-0.2, 0.3)), float(rs.uniform(-0.25, 0.25)), float(rs.choice([0.0, rs.uniform(-0.3, 0.3)])))
    sky = rs.randint(0, 256, size=(rs.randint(1, 6), rs.randint(1, 8), 4)).astype(np.uint8)
    wl = pt.scenes._finish(f"fuzz{seed}", sc, W, H, cam, rot, sky, int(rs.choice([1, 3, 8])), float(rs.choice([1, 4, 8, 12.5])),
                           blur=float(rs.choice([0.0, 0.001, 0.03])), auto_focus=float(rs.choice([0.0, 1.0])), focal_distance=float(rs.uniform(1.0, 4.0)))
    wl.textures = textures
    return wl


def _nested_shells_workload(pt, W=64, H=48, shells=11):
    """concentric transmissive shells, each with its own Ni: a ray towards the centre pushes one refraction index per shell
    (frag.glsl:833-836) — the index stack fills past its register-held slots 0-3, up to the ten the shader has (:136-158)"""
    S = pt.scenes
    sc = S._new_scene()
    S._cornell_materials(sc)
    for k in range(shells):
        sc.addMaterial(f"shell{k}")
        sc.setLastMtl("Tr", 1.0); sc.setLastMtl("Ni", 1.05 + 0.07 * k); sc.setLastMtl("Pr", 1)
        sc.setLastMtl("Tf", (0.02 * (k % 3), 0.01, 0.03)); sc.setLastMtl("Density", 0.5)
    o = S.Obj()
    S._cornell_room(o, boxes=False)
    for k in range(shells):
        o.group(f"shell{k}"); o.usemtl(f"shell{k}")
        o.mesh(*S.icosphere(1, (0.0, 0.55, 0.0), 0.52 - 0.04 * k))
    sc.addObjectText(o.text(), 0, parentDirectory="")
    return S._finish("shells", sc, W, H, S.CORNELL_CAM, S.CORNELL_ROT, (40, 50, 70), 4, 40)


@pytest.mark.parametrize("shells", [5, 8, 11])
def test_nested_transmissive_shells(pt, oracle, renderer_mod, shells):
    """index-stack slots 4-9 and the silent drop at ten entries, on both encodings of the stack in the path state: dictionary codes of 3 bits
    (5 shells: 8 distinct values with 0.0 and 1.0029) and of 8 bits (8 and 11 shells; forced for 5)"""
    wl = _nested_shells_workload(pt, shells=shells)
    got, ref, cnt, ocnt = render_both(pt, oracle, renderer_mod, wl, 3)
    assert_same(got, ref, cnt, ocnt)
    if shells == 5:
        got8, _, _, _ = render_both(pt, oracle, renderer_mod, wl, 3, index_stack_8bit=1)
        assert_same(got8, ref)
    r = renderer_mod.Renderer(wl.W, wl.H, devices=[0, 0])          # overlapped batches on the two-stream group
    seeds = seeds_for(pt, 1, 4)
    r.load_workload(wl); r.reset_frame()
    r.render_batch_async(1, seeds[:1]); r.render_batch_async(2, seeds[1:])
    got = r.read_frame(); r.close()
    ref, _ = oracle.render_frames(oracle.Scene.from_workload(wl), wl.W, wl.H, 1, 4, seeds, nthreads=8)
    assert_same(got, ref)


@pytest.mark.parametrize("shells", [5, 11])
def test_more_refraction_indices_than_the_dictionary_holds(pt, oracle, renderer_mod, shells):
    """The shader's refraction-index stack holds any float (frag.glsl:136-158, :834).  The path state normally carries it as dictionary codes (3 or 8 bits per slot);
    a scene with more than 254 distinct Ni among its materials — here the nested transmissive shells plus 300 materials that only widen the dictionary — switches to
    the state variant that carries the ten floats themselves (k_shade<32>), and the same variant forced onto the plain scene must give the same bits."""
    wl = _nested_shells_workload(pt, shells=shells)
    got32, ref, cnt, ocnt = render_both(pt, oracle, renderer_mod, wl, 3, index_stack_8bit=2)
    assert_same(got32, ref, cnt, ocnt)
    b = dict(wl.buffers)
    mtl = np.asarray(b[14], np.float32); me = int(mtl[0])
    extra = np.tile(mtl[1:1 + me], 300).reshape(300, me).copy()
    extra[:, 16] = 2.0 + 0.001 * np.arange(300, dtype=np.float32)       # Ni (dispatch.java:272-324: slot 16 of the record)
    b[14] = np.concatenate([mtl, extra.reshape(-1)]).astype(np.float32)
    wide = pt.scenes.Workload("shells+300", wl.W, wl.H, b, wl.sky, wl.sample_res, wl.max_bounces, wl.info)
    got, ref2, cnt, ocnt = render_both(pt, oracle, renderer_mod, wide, 3)
    assert_same(got, ref2, cnt, ocnt)
    assert np.array_equal(ref, ref2, equal_nan=True)                  # (the extra materials are never hit)
    r = renderer_mod.Renderer(wl.W, wl.H, devices=[0, 0])              # overlapped batches on the two-stream group, the pool growing under them
    seeds = seeds_for(pt, 1, 4)
    r.load_workload(wide); r.reset_frame()
    r.render_batch_async(1, seeds[:1]); r.render_batch_async(2, seeds[1:])
    got = r.read_frame(); r.close()
    ref4, _ = oracle.render_frames(oracle.Scene.from_workload(wide), wl.W, wl.H, 1, 4, seeds, nthreads=8)
    assert_same(got, ref4)


@pytest.mark.parametrize("seed", list(range(1, 25)))
def test_random_scenes(pt, oracle, renderer_mod, seed):
    wl = _random_workload(pt, seed, ellipsoid_maps=12 < seed <= 18, many_groups=seed > 18)
    got, ref, cnt, ocnt = render_both(pt, oracle, renderer_mod, wl, 3, index_stack_8bit=seed % 3)      # the three encodings of the index stack in the path state (3-bit / 8-bit codes, floats)
    assert_same(got, ref, cnt, ocnt)
    direct = wl.with_params(RAYTRACING=0)
    got, ref, cnt, ocnt = render_both(pt, oracle, renderer_mod, direct, 2, extend_mode=(2 if seed % 3 == 0 else seed % 2))      # (2: thickness probes on the hand-written kernel)
    assert_same(got, ref, cnt, ocnt)


@pytest.mark.parametrize("name,W,H", [("C3", 96, 54), ("C1", 64, 64), ("C4", 64, 36), ("T1", 48, 27)])
def test_debug_heatmap_mode(pt, oracle, renderer_mod, name, W, H):
    """DEBUG != 0 (frag.glsl:916-918, debugRayScene :539-547): the traversal heat-map, whole and tile-sharded, in any batch split"""
    wl = pt.scenes.build(name, W, H).with_params(DEBUG=1)
    got, ref, _, _ = render_both(pt, oracle, renderer_mod, wl, 3, count_stats=False)
    assert_same(got, ref)
    assert got[..., 2].min() > 0 and np.all(got[..., 1] == 0) and np.all(got[..., 3] == 3)
    acc = np.zeros_like(got)
    for rank in range(2):
        rr = renderer_mod.Renderer(W, H, shard_rank=rank, shard_count=2)
        rr.load_workload(wl); rr.reset_frame(); rr.render(1, 5); rr.render_batch(2, [6, 7])
        acc += rr.read_frame(); rr.close()
    assert np.array_equal(acc, ref)


def test_mixed_synchronous_and_overlapped_calls(pt, oracle, renderer_mod):
    """any interleaving of the synchronous and the overlapped entry points gives the frames of the equivalent synchronous sequence"""
    W, H = 80, 48
    wl = pt.scenes.build("C3", W, H)
    sc = oracle.Scene.from_workload(wl)
    s = [101, 202, 303, 404, 505, 606, 707, 808]
    r = renderer_mod.Renderer(W, H)
    r.set_option("path_slots", 1536)
    r.load_workload(wl)
    r.reset_frame()
    r.render_batch_async(1, s[0:2])                 # in flight ...
    r.render(3, s[2])                               # ... a synchronous frame joins the stream and completes everything
    r.render_batch_async(4, s[3:5])
    a = r.read_frame().copy()                       # completes the batch in flight
    ref = np.zeros((H, W, 4), np.float32)
    oracle.render_frames(sc, W, H, 1, 5, s[0:5], frame=ref, nthreads=8)
    assert np.array_equal(a, ref)
    r.render_batch_async(6, s[5:7])
    r.reset_frame()                                 # finishes what is in flight, then clears: the image restarts
    r.render_batch_async(1, s[7:8])
    r.next_image()                                  # that batch still lands in the image it was submitted for ...
    r.render_batch(1, s[0:1])                       # ... while this one fills the new image
    b = r.read_frame().copy()
    ref1 = np.zeros((H, W, 4), np.float32); oracle.render_frames(sc, W, H, 1, 1, s[0:1], frame=ref1, nthreads=8)
    assert np.array_equal(b, ref1)
    import torch
    from pathtracer_0_amd import shard
    prev = shard.frame_tensor(r, torch.device("cuda", 0), age=1).cpu().numpy().reshape(H, W, 4)
    ref2 = np.zeros((H, W, 4), np.float32); oracle.render_frames(sc, W, H, 1, 1, s[7:8], frame=ref2, nthreads=8)
    assert np.array_equal(prev, ref2)
    r.set_option("count_stats", 1)                  # an option change in the middle of nothing in flight
    r.render_batch_async(2, s[1:2]); r.synchronize()
    c = r.read_frame().copy()
    oracle.render_frames(sc, W, H, 2, 1, s[1:2], frame=ref1, nthreads=8)
    assert np.array_equal(c, ref1)
    r.close()


def test_gpu_bvh_builder_deep_unbalanced_tree(pt):
    """triangles on an exponential ladder x = 2^-3i: every split peels a few off, the tree runs into the builder's depth limit of 256
    (MAX_BVH_BRANCHES, dispatch.java:45) — the level loop, the per-thread subtree builder and the depth cut-off all have to agree with the CPU"""
    lines = ["o ladder", "vn 0 0 1"]
    k = 0
    for i in range(330):
        x, h = 2.0 ** (-3 * i), 2.0 ** (-3 * i - 3)
        for (dx, dy) in ((0.0, 0.0), (h, 0.0), (0.0, h)):
            lines.append("v %.17g %.17g 0" % (x + dx, dy))
        lines.append("f %d//1 %d//1 %d//1" % (k + 1, k + 2, k + 3))
        k += 3
    text = "\n".join(lines) + "\n"
    out = []
    for gpu in (False, True):
        sc = pt.hostlib.Scene(); sc.addMaterial("m")
        if gpu:
            sc.use_gpu_bvh_builder(0)
        sc.addObjectText(text, 0)
        out.append((sc.pack(), {k2: sc.count(k2) for k2 in ("nodes", "max_depth", "max_leaf", "leaf_indices")}))
    assert out[0][1] == out[1][1]
    assert out[0][1]["max_depth"] > 64
    for b in (3, 10, 11, 12, 13):
        assert np.array_equal(out[0][0][b].view(np.uint32), out[1][0][b].view(np.uint32)), b
