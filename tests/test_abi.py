"""CPU: the C-ABI libraries load and export every symbol include/*.h declares (no compute without a GPU)."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(pts?_[a-z_0-9]+)\s*\(", txt)))


def test_host_library_exports_every_declared_symbol(pt):
    from pathtracer_0_amd import build
    lib = ctypes.CDLL(build.build_host())
    names = declared("pt_scene.h")
    assert len(names) >= 12
    for n in names:
        assert hasattr(lib, n), n


def test_hip_library_exports_every_declared_symbol(pt):
    from pathtracer_0_amd import build
    lib = ctypes.CDLL(build.build_hip())       # hipcc cross-compiles gfx950 without a GPU
    names = declared("pt_api.h")
    assert len(names) >= 24 and "pt_create_multi" in names and "pt_gather_image" in names
    for n in names:
        assert hasattr(lib, n), n
    dbg = declared("pt_debug.h")                # tuning knobs / timers / parity probes live apart from the boundary header
    assert "pt_set_option" in dbg and "pt_set_option" not in names
    for n in dbg:
        assert hasattr(lib, n), n


def test_no_cpu_fallback_without_device(pt):
    """On a box without a GPU the product path must fail loudly, not fall back."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from pathtracer_0_amd import renderer
    with pytest.raises(renderer.PtError) as e:
        renderer.Renderer(64, 48)
    assert e.value.code == -2
    with pytest.raises(renderer.PtError) as e:
        renderer.Renderer(64, 48, devices=[0, 1])          # the multi-GPU context has no fallback either
    assert e.value.code == -2


def test_shard_maps_partition_the_image(pt):
    from pathtracer_0_amd import renderer
    import numpy as np
    for (W, H, n) in [(96, 40, 2), (100, 37, 3), (64, 64, 8), (1920, 1080, 8)]:
        slots = renderer.shard_slots(W, H, n)
        seen = np.zeros(W * H, int)
        for r in range(n):
            m = renderer.shard_map(W, H, r, n)
            assert len(m) == slots and slots % 256 == 0
            v = m[m >= 0]
            seen[v] += 1
            assert np.all(m[len(v):] == -1)
        assert np.all(seen == 1)
    assert np.array_equal(renderer.shard_map(8, 4, 0, 1), np.arange(32))


def _build_c_client(tmp_path):
    import subprocess
    exe = str(tmp_path / "abi_client")
    src = os.path.join(ROOT, "tests", "c", "abi_client.c")
    out = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I" + os.path.join(ROOT, "include"), "-o", exe, src, "-ldl"],
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    return exe


def test_headers_are_valid_c_and_the_c_client_builds(tmp_path):
    """include/pt_api.h and include/pt_scene.h compile as strict C99 in a plain-C client (what a JNI shim or any compiled host includes)"""
    _build_c_client(tmp_path)


@pytest.mark.gpu
def test_c_client_renders_what_the_python_wrappers_render(pt, renderer_mod, tmp_path):
    """the same scene through the C ABIs from plain C (dlopen, no Python in the process) and through hostlib/renderer: same bits"""
    import subprocess
    import numpy as np
    exe = _build_c_client(tmp_path)
    W, H, frames = 96, 54, 3
    out = subprocess.run([exe, os.path.join(ROOT, "pathtracer-0_amd"), str(W), str(H), str(frames)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ABI_CLIENT_OK" in out.stdout, out.stdout + out.stderr
    fnv = out.stdout.split("fnv1a ")[1].split()[0]
    sc = pt.hostlib.Scene()
    sc.addMaterial("default"); sc.setLastMtl("Kd", (0.8, 0.8, 0.8)); sc.setLastMtl("Pr", 1)
    sc.addMaterial("lamp"); sc.setLastMtl("Ke", (12, 12, 12))
    sc.addMaterial("metal"); sc.setLastMtl("Pm", 1); sc.setLastMtl("Pr", 0.2)
    quad = ("o floor\nvn 0 1 0\nv -2 0 -2\nv 2 0 -2\nv 2 0 2\nv -2 0 2\nf 1//1 2//1 3//1\nf 1//1 3//1 4//1\n"
            "o lamp\nusemtl lamp\nvn 0 -1 0\nv -0.5 2 -0.5\nv 0.5 2 -0.5\nv 0.5 2 0.5\nv -0.5 2 0.5\nf 5//2 7//2 6//2\nf 5//2 8//2 7//2\n")
    sc.addObjectText(quad, 0, parentDirectory="")
    sc.addEllipsoid((0.0, 0.5, 0.0), 1.0, 0.0, 0.5, 2)
    wl = pt.scenes._finish("c_client", sc, W, H, (0.0, 1.0, -3.0), (0.0, 0.0, 0.0), (150, 180, 230), 8, 4)
    r = renderer_mod.Renderer(W, H)
    r.load_workload(wl); r.reset_frame()
    for f in range(1, frames + 1):
        r.render(f, (1234 + 7919 * f) % 10000)
    img = r.read_frame(); r.close()
    h = 1469598103934665603
    for byte in img.tobytes():
        h = ((h ^ byte) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    assert "%016x" % h == fnv
    assert np.all(img[..., 3] == frames)


@pytest.mark.gpu
def test_c_host_runs_the_bench_schedule(pt, renderer_mod, tmp_path):
    """tests/c/bench_client.c: bench.py's schedule (pt_next_image, asynchronous batches, the image gathered two steps later, two streams behind ONE context)
    driven from plain C on the HIP runtime the library links by itself — the process a Java / C host is (scripts/c_host_bench.py times it at full size for
    INTEGRATION.md section 3).  The last image equals the same frames rendered through the Python wrappers."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("c_host_bench", os.path.join(ROOT, "scripts", "c_host_bench.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    W, H, fps = 192, 108, 3
    d = json.loads(m.run("C3", steps=4, warmup=1, streams=2, fps=fps, W=W, H=H, workdir=str(tmp_path)))
    assert d["value"] > 0 and d["count"] == fps and "libamdhip64" in d["hip_runtime"] and "torch" not in d["hip_runtime"]
    wl = pt.scenes.build("C3", W, H)
    r = renderer_mod.Renderer(W, H)
    r.load_workload(wl); r.reset_frame(); r.render_batch(1, [pt.scenes.frame_seed(f) for f in range(1, 1 + fps)])
    img = r.read_frame(); r.close()
    h = 1469598103934665603
    for byte in img.tobytes():
        h = ((h ^ byte) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    assert "%016x" % h == d["fnv1a"]


def test_gather_layout_roundtrip(pt):
    """The bookkeeping of the one collective: every rank's packed accumulator (pt_shard_map order, padded to pt_shard_slots) laid
    rank-major — what ncclGather delivers on the root — and scattered through the concatenated maps (k_unshard's contract) is the image."""
    from pathtracer_0_amd import renderer
    import numpy as np
    rng = np.random.default_rng(5)
    for (W, H, n) in [(96, 40, 2), (100, 37, 3), (33, 9, 4), (1920, 1080, 8)]:
        full = rng.random((W * H, 4), dtype=np.float32)
        slots = renderer.shard_slots(W, H, n)
        maps = np.concatenate([renderer.shard_map(W, H, r, n) for r in range(n)])
        gathered = np.full((n * slots, 4), np.nan, np.float32)         # padding slots carry garbage and must never be read
        for r in range(n):
            m = maps[r * slots:(r + 1) * slots]
            gathered[r * slots:(r + 1) * slots][m >= 0] = full[m[m >= 0]]
        out = np.zeros_like(full)
        out[maps[maps >= 0]] = gathered[maps >= 0]
        assert np.array_equal(out, full)


def test_jni_shim_covers_the_boundary():
    """pathtracer-0_amd/java: every native method of Main.PtNative has its JNI function in pt_jni.c, every C-ABI call the shim makes is
    declared in include/pt_api.h, and the shim reaches every entry point a Java host needs (no JDK here: a text-level consistency check)"""
    java = open(os.path.join(ROOT, "pathtracer-0_amd", "java", "Main", "PtNative.java")).read()
    jni = open(os.path.join(ROOT, "pathtracer-0_amd", "java", "pt_jni.c")).read()
    natives = re.findall(r"public static native [\w\[\]]+ (\w+)\(", java)
    assert len(natives) >= 21
    for n in natives:
        assert f"Java_Main_PtNative_{n}(" in jni, n
    assert len(re.findall(r"JNIEXPORT", jni)) == len(natives)
    api = set(declared("pt_api.h"))
    called = set(re.findall(r"\b(pt_[a-z_]+)\(", jni))
    assert called <= api, called - api
    for need in ("pt_create", "pt_create_multi", "pt_create_multi_part", "pt_stream_wait", "pt_destroy", "pt_set_buffer", "pt_set_texture", "pt_reset_frame", "pt_render", "pt_render_batch", "pt_render_batch_async",
                 "pt_next_image", "pt_finish_image", "pt_image_device", "pt_gather_image", "pt_synchronize", "pt_read_frame", "pt_write_frame", "pt_read_display", "pt_get_counters",
                 "pt_reset_counters", "pt_last_error"):
        assert need in called, need


def test_jni_shim_type_checks():
    """N1 as far as a box without a JDK allows: pt_jni.c compiled (-fsyntax-only -Wall -Wextra -Werror) against tests/c/jni_standin/jni.h, a
    hand-written stand-in that declares the JNI types and the eleven JNIEnv functions the shim calls with the prototypes of the JNI
    specification, and every Java_Main_PtNative_* definition compared with the `native` declaration of PtNative.java it implements (return
    type and every parameter type, in order).  This pins nothing about a real JVM — nothing is linked or run, and the stand-in is the
    builder's reading of the specification; it proves that the file is valid C, that its calls match those prototypes and that the two
    sides of the JNI boundary agree.  N1 itself (the reference's Main driving the library) stays open: no JDK here or on the GPU boxes."""
    jdir = os.path.join(ROOT, "pathtracer-0_amd", "java")
    cmd = ["gcc", "-std=c11", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-Wno-unused-parameter", "-I" + os.path.join(ROOT, "tests", "c", "jni_standin"),
           "-I" + os.path.join(ROOT, "include"), os.path.join(jdir, "pt_jni.c")]
    out = subprocess.run(cmd, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    java = open(os.path.join(jdir, "Main", "PtNative.java")).read()
    jni = open(os.path.join(jdir, "pt_jni.c")).read()
    jtype = {"int": "jint", "long": "jlong", "boolean": "jboolean", "void": "void", "int[]": "jintArray", "long[]": "jlongArray", "Buffer": "jobject", "String": "jstring"}
    decl = {m.group(2): (m.group(1), [a.split()[0] for a in m.group(3).split(",") if a.strip()])
            for m in re.finditer(r"public static native ([\w\[\]]+) (\w+)\(([^)]*)\)", java)}
    defs = {m.group(2): (m.group(1), [a.rsplit(None, 1)[0].strip() for a in m.group(3).split(",")])
            for m in re.finditer(r"JNIEXPORT (\w+) JNICALL Java_Main_PtNative_(\w+)\(([^)]*)\)", jni)}
    assert set(decl) == set(defs) and len(decl) >= 21
    for name, (ret, args) in decl.items():
        cret, cargs = defs[name]
        assert cret == jtype[ret], (name, cret, ret)
        assert cargs[:2] == ["JNIEnv*", "jclass"], (name, cargs[:2])          # static native methods: (JNIEnv*, jclass, ...)
        assert cargs[2:] == [jtype[a] for a in args], (name, cargs[2:], args)


def test_c_client_names_every_boundary_symbol():
    """tests/c/abi_client.c drives every entry point of include/pt_api.h (the GPU test runs it; this one keeps the list complete)"""
    src = open(os.path.join(ROOT, "tests", "c", "abi_client.c")).read()
    for n in declared("pt_api.h"):
        assert f"SYM(hip, {n})" in src, n


def _build_jni_harness(tmp_path):
    """pt_jni.c + tests/c/jni_harness.c (a JNIEnv made of plain C functions) -> an executable linked against libpt_hip.so"""
    import subprocess
    exe = str(tmp_path / "jni_harness")
    lib = os.path.join(ROOT, "pathtracer-0_amd")
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-Wno-unused-parameter", "-I" + os.path.join(ROOT, "tests", "c", "jni_standin"), "-I" + os.path.join(ROOT, "include"),
           "-o", exe, os.path.join(lib, "java", "pt_jni.c"), os.path.join(ROOT, "tests", "c", "jni_harness.c"), "-L" + lib, "-lpt_hip", "-Wl,-rpath," + lib]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    return exe


def test_jni_shim_links_into_the_jvm_free_harness(pt, tmp_path):
    """the shim, compiled for real (not -fsyntax-only) against the stand-in header, links with the harness and libpt_hip.so: every native it defines resolves"""
    from pathtracer_0_amd import build
    build.build_hip()
    _build_jni_harness(tmp_path)


@pytest.mark.gpu
@pytest.mark.parametrize("two_streams", [0, 1])
def test_jni_shim_runs_without_a_jvm(pt, renderer_mod, tmp_path, two_streams):
    """N1 one step further than a type check: pt_jni.c EXECUTES — a JNIEnv of plain C functions (tests/c/jni_harness.c: a direct buffer is (address, capacity),
    an int[] is (length, data), a pending exception is (class, message)) drives the Java_Main_PtNative_* natives in the order the reference's Main would:
    uploads from direct buffers, one draw, a batch, draws left in flight, glFinish, glReadPixels, the screenshot, counters, the image ring, the error paths
    (heap buffer -> IllegalArgumentException, library error -> RuntimeException with pt_last_error()).  FRAME and the display bytes must equal what the
    ctypes face renders.  It still pins nothing about a real JVM (no JDK here): the stand-in header is the builder's reading of the JNI specification."""
    import subprocess
    import numpy as np
    W, H = 96, 54
    wl = pt.scenes.build("C3", W, H)
    d = tmp_path / "in"; d.mkdir()
    for b in (0, 1, 2, 3, 4, 5, 7, 10, 11, 12, 13, 14):
        np.ascontiguousarray(wl.buffers[b]).tofile(str(d / f"binding_{b}.bin"))
    sky = np.ascontiguousarray(wl.sky, dtype=np.uint8)
    sky.tofile(str(d / "sky.bin"))
    exe = _build_jni_harness(tmp_path)
    out = subprocess.run([exe, str(d), str(W), str(H), str(sky.shape[1]), str(sky.shape[0]), str(two_streams)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "JNI_HARNESS_OK" in out.stdout, out.stdout + out.stderr
    assert "setBuffer(heap buffer) -> java/lang/IllegalArgumentException" in out.stdout and "render without a scene -> java/lang/RuntimeException" in out.stdout
    seeds = [9153, 7072, 4991, 2910, 829, 8748]
    r = renderer_mod.Renderer(W, H)
    r.load_workload(wl); r.reset_frame(); r.reset_counters()
    r.render_batch(1, seeds)
    ref = r.read_frame().copy(); disp = r.read_display(6, java_bytes=True); cnt = r.counters()
    r.render(7, 6667); ref7 = r.read_frame().copy(); r.close()
    got7 = np.fromfile(str(d / "frame7.bin"), dtype=np.float32).reshape(H, W, 4)
    assert np.array_equal(got7, ref7, equal_nan=True) and np.all(got7[..., 3] == 7)      # writeFrame(frames 1..6) + frame 7 == seven frames in one go
    got = np.fromfile(str(d / "frame.bin"), dtype=np.float32).reshape(H, W, 4)
    assert np.array_equal(got, ref, equal_nan=True) and np.all(got[..., 3] == 6)
    assert np.array_equal(np.fromfile(str(d / "display.bin"), dtype=np.uint8).reshape(disp.shape), disp)
    line = [l for l in out.stdout.splitlines() if l.startswith("counters:")][0].split()[1:]
    assert int(line[4]) == cnt["samples"] and int(line[0]) == cnt["segments"]            # PT_CNT_SEGMENTS, PT_CNT_SAMPLES through getCounters' long[]
    assert os.path.getsize(str(d / "shot.png")) > 100
