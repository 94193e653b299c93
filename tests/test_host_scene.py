"""CPU: host-side scene producers (C++ mirror of the reference's Java scene DSL / BVH builder / packers)."""
import os

import numpy as np
import pytest

CUBES = """o Cube
v 1 1 -1
v 1 -1 -1
v 1 1 1
v 1 -1 1
v -1 1 -1
v -1 -1 -1
v -1 1 1
v -1 -1 1
f 5 3 1
f 3 8 4
f 7 6 8
f 2 8 6
f 1 4 2
f 5 2 6
f 5 7 3
f 3 7 8
f 7 5 6
f 2 4 8
f 1 3 4
f 5 1 2
"""


def _check_tree(b, n_tris_expected, n_obj_expected):
    """K8: objIndices = [count, roots], BVHtree sorted by id, ids are DFS pre-order, leaf ranges contiguous and a partition"""
    tree = b[11].reshape(-1, 3); data = b[10].reshape(-1, 8); leaf = b[12]; obj = b[13]
    assert obj[0] == n_obj_expected == len(obj) - 1
    assert np.array_equal(tree[:, 0], np.arange(len(tree)))
    seen_leaf = np.zeros(len(leaf), bool)
    next_id = [0]

    def walk(n, lo, hi):
        assert n == next_id[0]; next_id[0] += 1              # construction order == DFS pre-order (Q-11)
        l, r = tree[n, 1], tree[n, 2]
        assert np.all(data[n, :3] >= lo - 1e-6) and np.all(data[n, 3:6] <= hi + 1e-6)
        if l == -1 and r == -1:
            s, e = int(data[n, 6]), int(data[n, 7])
            assert e > s and not seen_leaf[s:e].any()
            seen_leaf[s:e] = True
            tri = b[3].reshape(-1, 40)[leaf[s:e]]
            pts = tri[:, [0, 1, 2, 4, 5, 6, 8, 9, 10]].reshape(-1, 3)
            assert np.allclose(pts.min(0), data[n, :3]) and np.allclose(pts.max(0), data[n, 3:6])
            return
        assert l >= 0 and r >= 0 and data[n, 6] == 0 and data[n, 7] == 0
        walk(l, data[n, :3], data[n, 3:6]); walk(r, data[n, :3], data[n, 3:6])

    for root in obj[1:]:
        walk(int(root), np.full(3, -np.inf), np.full(3, np.inf))
    assert next_id[0] == len(tree) and seen_leaf.all()
    assert sorted(leaf.tolist()) == list(range(n_tris_expected))


def test_k8_flatten_two_objects(pt):
    sc = pt.hostlib.Scene()
    sc.addMaterial("default")
    two = CUBES + CUBES.replace("o Cube", "o Cube.002").replace("f 5 3 1", "f 13 11 9")
    # second cube reuses the global vertex numbering (+8)
    lines = []
    for ln in CUBES.splitlines():
        if ln.startswith("f "):
            lines.append("f " + " ".join(str(int(t) + 8) for t in ln.split()[1:]))
        elif ln.startswith("v "):
            x, y, z = map(float, ln.split()[1:]); lines.append(f"v {x + 5} {y} {z}")
        else:
            lines.append(ln.replace("Cube", "Cube.002"))
    sc.addObjectText(CUBES + "\n".join(lines) + "\n", 0)
    b = sc.pack()
    assert sc.count("triangles") == 24 and sc.count("objects") == 2
    _check_tree(b, 24, 2)
    # no vn -> NaN normals are uploaded as they are, vt sentinel 69.420 (Q-7), material id in slot 36
    t = b[3].reshape(-1, 40)
    assert np.isnan(t[:, 12:15]).all() and np.all(t[:, 24] == np.float32(69.420)) and np.all(t[:, 36] == 0)
    assert b[14][0] == 48.0 and len(b[14]) == 49
    assert b[5].tolist() == [0.0] and b[7].tolist() == [0.0]


@pytest.mark.skipif(not os.path.exists("/root/reference/src/objs/table - Copy.obj"), reason="reference checkout not present (GPU box)")
def test_k8_shipped_obj(pt):
    sc = pt.hostlib.Scene()
    sc.addMaterial("default")
    sc.addObject("/root/reference/src/objs/table - Copy.obj", 0)
    b = sc.pack()
    assert sc.count("triangles") == 60 and sc.count("objects") == 5
    _check_tree(b, 60, 5)


def test_material_defaults_and_packing_order(pt):
    sc = pt.hostlib.Scene()
    sc.addMaterial("default"); sc.setLastMtl("Kd", (0.8, 0.8, 0.8)); sc.setLastMtl("Pr", 1)
    sc.addMaterial("test")
    for k, v in dict(Kd=(0.8, 0.45, 0.5), Ks=(0.5, 0.5, 0.5), Ni=1.45, Pr=1, Pc=0.0, Pcr=0.0, Tr=0.7, subsurface=0, subsurfaceColor=(0.45, 0.8, 0.5),
                     subsurfaceRadius=(1, 1, 1), Density=0.1).items():
        sc.setLastMtl(k, v)
    m = sc.pack()[14]
    d = m[1:49]          # dispatch.java:1514-1550 defaults in the :272-324 order
    exp = [0, 0, 0, .8, .8, .8, .5, .5, .5, 10, 0, 0, 0, 0, 0, 1, 0, 0, 0, 1, 0, -1, -1, -1, 0, 1, 0, 0, 0, 0, 0, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 0, 0, 0, 0, 0, 0, 0]
    assert np.allclose(d, np.array(exp, np.float32))
    t = m[49:97]
    assert np.allclose(t[3:6], (0.8, 0.45, 0.5)) and t[11] == np.float32(0.7) and t[15] == np.float32(1.45) and t[19] == np.float32(0.1)
    assert np.allclose(t[42:45], (0.45, 0.8, 0.5)) and np.allclose(t[45:48], (1, 1, 1))
    with pytest.raises(RuntimeError, match="Not a valid property"):
        sc.setLastMtl("nope", 1)
    with pytest.raises(RuntimeError):
        sc.setLastMtl("Kd", 1.0)


def test_transform_order_and_normals(pt):
    """v -> mult(scale).rotate(rot).add(shift); vn -> mult(scale).rotate(rot) then normalised (Q-10)"""
    sc = pt.hostlib.Scene(); sc.addMaterial("m")
    obj = "o q\nv 1 0 0\nv 0 1 0\nv 0 0 1\nv 1 1 1\nvn 1 0 0\nf 1//1 2//1 3//1\nf 2//1 3//1 4//1\n"
    sc.addObjectText(obj, 0, scale=(2, 3, 4), shift=(10, 20, 30), rot=(0, 0, np.pi / 2))
    t = sc.pack()[3].reshape(-1, 40)[0]
    assert np.allclose(t[0:3], (10, 22, 30), atol=1e-6)        # (2,0,0) rotated 90deg about z -> (0,2,0), + shift
    assert np.allclose(t[4:7], (7, 20, 30), atol=1e-6)         # (0,3,0) -> (-3,0,0)
    assert np.allclose(t[12:15], (0, 1, 0), atol=1e-6)


def test_unsplittable_root_is_an_error(pt):
    sc = pt.hostlib.Scene(); sc.addMaterial("m")
    with pytest.raises(RuntimeError, match="Q-16"):
        sc.addObjectText("o one\nv 0 0 0\nv 1 0 0\nv 0 1 0\nvn 0 0 1\nf 1//1 2//1 3//1\n", 0)


def test_usemtl_name_concatenation(pt):
    sc = pt.hostlib.Scene(); sc.addMaterial("a"); sc.addMaterial("bnull"); sc.addMaterial("c/dir")
    quad = "v 0 0 {z}\nv 1 0 {z}\nv 1 1 {z}\nv 0 1 {z}\nvn 0 0 1\n"
    sc.addObjectText("o q\nusemtl b\n" + quad.format(z=0) + "f 1//1 2//1 3//1\nf 1//1 3//1 4//1\n", 0)            # parentDirectory null -> "bnull"
    sc.addObjectText("o q\nusemtl c\n" + quad.format(z=1) + "f 1//1 2//1 3//1\nf 1//1 3//1 4//1\n", 0, parentDirectory="/dir")
    t = sc.pack()[3].reshape(-1, 40)
    assert t[0, 36] == 1 and t[2, 36] == 2


def test_workloads_build_and_are_deterministic(pt):
    for name, (W, H) in dict(C1=(64, 64), C2=(64, 36), C3=(64, 36)).items():
        a, b = pt.scenes.build(name, W, H), pt.scenes.build(name, W, H)
        for k in a.buffers:
            assert np.array_equal(a.buffers[k], b.buffers[k], equal_nan=True)
        assert a.buffers[4][2] == W and int(a.buffers[4][2] * a.buffers[4][3]) == H
    assert pt.scenes.build("C2", 64, 36).info["triangles"] == 36
    assert pt.scenes.build("C3", 64, 36).info["triangles"] == 12 + 2 * 1280
    assert [pt.scenes.frame_seed(f) for f in (1, 2, 3)] == [9153, 7072, 4991]


# ---- material.parseMtls + scene.addObject on a directory (dispatch.java:869-882, :1319-1512, :1552-1575)
MTL = """# comment
newmtl wood
Ka 0.1 0.2 0.3
Kd 0.5 0.6 0.7
Ks 0.25 0.25 0.25
Ns 96
d 0.75
Ni 1.45
illum 2
map_Kd tex/wood.png
map_Bump tex/bump.png
Pr 0.4
subsurfaceColor 1 0.5 0.25

newmtl lamp
Ke 3 4 0
Tr 0.25
map_Ke tex/wood.png
refl tex/rough.png
"""

QUADS = """o floor
usemtl wood
v -1 0 -1
v 1 0 -1
v 1 0 1
v -1 0 1
vt 0 0
vt 1 0
vt 1 1
vt 0 1
vn 0 1 0
f 1/1/1 2/2/1 3/3/1
f 1/1/1 3/3/1 4/4/1
o panel
usemtl lamp
v -1 2 -1
v 1 2 -1
v 1 2 1
v -1 2 1
f 5/1/1 6/2/1 7/3/1
f 5/1/1 7/3/1 8/4/1
"""


def _asset_dir(tmp_path):
    from PIL import Image
    d = tmp_path / "asset"; (d / "tex").mkdir(parents=True)
    (d / "a.mtl").write_text(MTL); (d / "b.OBJ").write_text(QUADS)
    rng = np.random.default_rng(5)
    for name, mode in (("wood", "RGB"), ("bump", "RGBA"), ("rough", "L")):
        ch = {"RGB": 3, "RGBA": 4, "L": 1}[mode]
        a = rng.integers(0, 256, (6, 5, ch), dtype=np.uint8)
        Image.fromarray(a[..., 0] if ch == 1 else a, mode).save(d / "tex" / f"{name}.png")
    return str(d)


def test_parse_mtls_and_directory_import(pt, tmp_path):
    d = _asset_dir(tmp_path)
    sc = pt.hostlib.Scene()
    assert sc.addTexture("sky.png", "skybox.png") == 0          # dispatch.java:221-222: the sky is texture 0
    sc.addMaterial("default")
    sc.addObject(d, 0)                                          # a directory: every .mtl, then every .obj (case-insensitive)
    names = [n for _, n in sc.textures()]
    assert names == ["skybox.png", "tex\\wood.png", "tex\\bump.png", "tex\\rough.png"]      # '/' -> '\\' (:1330); deduplicated by name
    assert [p for p, _ in sc.textures()][1] == d + "/tex/wood.png"
    b = sc.pack()
    assert b[14][0] == 48
    m = b[14][1:].reshape(-1, 48)
    assert m.shape[0] == 3
    wood, lamp = m[1], m[2]
    np.testing.assert_array_equal(wood[0:9], np.float32([0.1, 0.2, 0.3, 0.5, 0.6, 0.7, 0.25, 0.25, 0.25]))
    assert wood[9] == 96 and wood[10] == np.float32(0.75) and wood[11] == np.float32(0.25) and wood[15] == np.float32(1.45)   # d sets Tr = 1 - d
    assert wood[19] == 1 and wood[20] == 2                      # Density default, illum
    assert wood[22] == 1 and wood[36] == 2 and wood[21] == -1   # map_Kd, map_bump, map_Ka
    assert wood[25] == np.float32(0.4)
    np.testing.assert_array_equal(wood[42:45], np.float32([1, 0.5, 0.25]))
    assert lamp[19] == 5 and lamp[11] == np.float32(0.25) and lamp[10] == np.float32(0.75)     # Density = |Ke|; Tr sets d
    assert lamp[40] == 1 and lamp[32] == 3                      # map_Ke reuses wood.png by name; refl -> map_Pr
    t = b[3].reshape(-1, 40)
    assert t.shape[0] == 4 and list(t[:, 36]) == [1, 1, 2, 2]   # usemtl resolves <name><dir> (:924, :1328)
    tex = sc.load_textures()
    assert sorted(tex) == [1, 2, 3] and tex[1].shape == (6, 5, 4) and (tex[1][..., 3] == 255).all()
    assert (tex[3][..., 0] == tex[3][..., 1]).all()             # grey -> RGBA like stbi_load(..., 4)


def test_parse_mtls_errors(pt, tmp_path):
    d = _asset_dir(tmp_path)
    sc = pt.hostlib.Scene()
    with pytest.raises(RuntimeError, match="cannot open MTL"):
        sc.parseMtls(d + "/nope.mtl", d)
    (tmp_path / "bad.mtl").write_text("newmtl x\nmap_Kd missing.png\n")
    with pytest.raises(RuntimeError, match="cannot read texture"):
        sc.parseMtls(str(tmp_path / "bad.mtl"), str(tmp_path))
    (tmp_path / "bad2.mtl").write_text("newmtl x\nKd 0.5  0.5 0.5\n")       # double space: split(" ") yields an empty token
    with pytest.raises(RuntimeError, match="NumberFormatException"):
        sc.parseMtls(str(tmp_path / "bad2.mtl"), str(tmp_path))
    empty = tmp_path / "empty"; empty.mkdir()
    with pytest.raises(RuntimeError, match="no obj files"):
        sc.addObject(str(empty), 0)
