"""Render call of the reference behind the C ABI (include/pt_api.h), Python face.

`Renderer` plays the part of the GL state the reference's frame loop drives
(/root/reference/src/Main/dispatch.java:590-713): set_buffer == glBufferData/glBufferSubData on
an SSBO binding point, set_texture == texture upload, reset_frame == resetTexture (:732-735),
render(frame_count, seed) == glUniform1i x2 + glDrawArrays (:697-705), read_frame == glReadPixels
of the RGBA32F accumulation image.  There is no CPU fallback: constructing a Renderer without
the HIP library or without a gfx950 device raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

COUNTERS = ["segments", "nodes", "tritests", "hitupd", "samples", "boxtests", "iterations", "extend_launches"]
KERNELS = {"extend": 0, "shade": 1, "generate": 2, "accumulate": 3}
OPTIONS = {"path_slots": 0, "count_stats": 1, "lds_budget": 2, "none_min": 3, "extend_mode": 4, "extend_tpb": 5, "extend_cache_bytes": 6,
           "refill_min": 7, "extend_blocks_per_cu": 8, "inner_keep_eighths": 9, "bfs_nodes": 10, "stack_mode": 11,
           "query_asm_eligible": 12, "query_asm_launches_above": 13, "asm_loop": 14, "numeric_contract": 16, "asm_tpb": 17, "index_stack_8bit": 18, "asm_node_layout": 19, "asm_root_cull": 20, "cu_partition": 21}


def hip_runtimes_mapped():
    """real paths of every libamdhip64 mapped into this process (Linux)"""
    out = set()
    try:
        for line in open("/proc/self/maps"):
            p = line.rstrip("\n").split(" ")[-1]
            if "libamdhip64" in os.path.basename(p):
                out.add(os.path.realpath(p))
    except OSError:
        pass
    return sorted(out)


def hip_runtime_info():
    """{"path", "version"} of the HIP runtime this process runs on (hipRuntimeGetVersion of the mapped libamdhip64)"""
    rts = hip_runtimes_mapped()
    if not rts:
        return {"path": None, "version": None}
    v = C.c_int(0)
    try:
        C.CDLL(rts[0]).hipRuntimeGetVersion(C.byref(v))
    except (OSError, AttributeError):
        pass
    return {"path": rts[0], "version": v.value, "runtimes_mapped": len(rts)}


def _load_one_hip_runtime(path):
    """libpt_hip.so is linked against libamdhip64.so.7 (RUNPATH: the ROCm installation it was built with).  PyTorch ships its OWN
    libamdhip64.so with the same SONAME.  A process must run on ONE of them:
      * torch imported first  -> the dynamic loader resolves the library's libamdhip64.so.7 to torch's copy, already mapped: one runtime;
      * library loaded first  -> ROCm's runtime is mapped and initialises the GPU; a later `import torch` maps a SECOND runtime, which finds
        no device ("No HIP GPUs are available", profiles/r03_a_hip_runtime_probe.txt — the failure conftest.py used to paper over).
    So: a Python host that will also use torch (bench.py, the tests, shard.py) gets torch imported here, before the library; a process that
    ends up with two runtimes mapped is refused.  PT_NO_TORCH=1: never import torch (a torch-free host; it then runs on ROCm's runtime)."""
    import sys
    if "torch" not in sys.modules and os.environ.get("PT_NO_TORCH") != "1":
        try:
            import torch  # noqa: F401  (maps torch's HIP runtime first)
        except ImportError:
            pass
    L = C.CDLL(path)
    rts = hip_runtimes_mapped()
    if len(rts) > 1 and os.environ.get("PT_ALLOW_TWO_HIP_RUNTIMES") != "1":
        raise RuntimeError("two HIP runtimes are mapped into this process: " + ", ".join(rts) + ".  libpt_hip.so and PyTorch must share one: "
                           "import torch BEFORE the first pathtracer_0_amd.renderer call (or set PT_NO_TORCH=1 and do not import torch at all)")
    return L


def lib():
    global _LIB
    if _LIB is None:
        path = os.environ.get("PT_HIP_LIB") or os.path.join(_HERE, "libpt_hip.so")     # PT_HIP_LIB: A/B builds of the same ABI (tuning only)
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: the HIP extension is required (no fallback). Build it with __graft_entry__.build()")
        L = _load_one_hip_runtime(path)
        vp, ci, sz = C.c_void_p, C.c_int, C.c_size_t
        L.pt_last_error.restype = C.c_char_p
        L.pt_create.argtypes = [C.POINTER(vp), ci, ci, ci, ci, ci]
        L.pt_create_multi.argtypes = [C.POINTER(vp), C.POINTER(ci), ci, ci, ci]
        L.pt_create_multi_part.argtypes = [C.POINTER(vp), C.POINTER(ci), ci, ci, ci, ci, ci]
        L.pt_gather_image.argtypes = [vp, ci, C.POINTER(vp)]
        L.pt_destroy.argtypes = [vp]
        L.pt_set_buffer.argtypes = [vp, ci, vp, sz]
        L.pt_set_texture.argtypes = [vp, ci, ci, ci, vp]
        L.pt_reset_frame.argtypes = [vp]
        L.pt_render.argtypes = [vp, ci, ci]
        L.pt_render_batch.argtypes = [vp, ci, ci, vp]
        L.pt_render_batch_async.argtypes = [vp, ci, ci, vp]
        L.pt_next_image.argtypes = [vp]
        L.pt_finish_image.argtypes = [vp, ci]
        L.pt_image_device.argtypes = [vp, ci, C.POINTER(vp), C.POINTER(sz)]
        L.pt_synchronize.argtypes = [vp]
        L.pt_stream_wait.argtypes = [vp]
        L.pt_read_frame.argtypes = [vp, vp]
        if hasattr(L, "pt_write_frame"):                      # (absent in A/B builds of earlier rounds loaded through PT_HIP_LIB)
            L.pt_write_frame.argtypes = [vp, vp]
        L.pt_read_display.argtypes = [vp, ci, ci, vp]
        L.pt_save_png.argtypes = [vp, ci, ci, C.c_char_p]
        L.pt_frame_device.argtypes = [vp, C.POINTER(vp), C.POINTER(sz)]
        L.pt_shard_slots.argtypes = [ci, ci, ci, C.POINTER(sz)]
        L.pt_shard_map.argtypes = [ci, ci, ci, ci, vp, sz]
        L.pt_unshard.argtypes = [vp, vp, vp]
        L.pt_set_stream.argtypes = [vp, vp]
        L.pt_set_option.argtypes = [vp, ci, C.c_int64]
        L.pt_get_counters.argtypes = [vp, vp, ci]
        L.pt_reset_counters.argtypes = [vp]
        L.pt_kernel_time.argtypes = [vp, ci, C.POINTER(C.c_int64), C.POINTER(C.c_double)]
        L.pt_set_timing.argtypes = [vp, ci]
        L.pt_kernel_time_median.argtypes = [vp, ci, C.POINTER(C.c_double)]
        L.pt_debug_math.argtypes = [vp, ci, vp, vp, vp, sz]
        L.pt_debug_intersect.argtypes = [vp, vp, vp, vp, sz]
        _LIB = L
    return _LIB


class PtError(RuntimeError):
    """Error code + message of the C ABI (the reference throws RuntimeException at these points)."""

    def __init__(self, code, msg):
        super().__init__(f"[{code}] {msg}")
        self.code = code


def _check(rc):
    if rc != 0:
        raise PtError(rc, lib().pt_last_error().decode())


def shard_slots(W, H, count):
    n = C.c_size_t()
    _check(lib().pt_shard_slots(W, H, count, C.byref(n)))
    return n.value


def shard_map(W, H, rank, count):
    n = shard_slots(W, H, count)
    out = np.empty(n, dtype=np.int32)
    _check(lib().pt_shard_map(W, H, rank, count, out.ctypes.data, n))
    return out


class Renderer:
    """One render context.  `devices=[...]`: ONE context made of several wavefront streams (pt_create_multi): one entry per stream —
    several GPUs ([0, 1, ...]), several independent streams on one GPU ([0, 0]: their kernels overlap, 11-24 % faster than one: include/pt_api.h), or both
    ([0, 0, 1, 1]); the tile shards, the per-stream host threads and the gather of every image live inside the library.
    first_shard / total_shards: the group renders only shards first_shard.. of total_shards (one process per GPU; pt_create_multi_part)."""

    def __init__(self, W, H, device=0, shard_rank=0, shard_count=1, devices=None, first_shard=0, total_shards=None):
        self._L = lib()
        self._h = C.c_void_p()
        self.W, self.H, self.shard_rank, self.shard_count = W, H, shard_rank, shard_count
        self.devices = None if devices is None else [int(d) for d in devices]
        if self.devices is not None:
            assert shard_count == 1 and shard_rank == 0, "a multi-stream context shards by itself"
            arr = (C.c_int * len(self.devices))(*self.devices)
            self.first_shard, self.total_shards = int(first_shard), int(total_shards if total_shards is not None else len(self.devices))
            _check(self._L.pt_create_multi_part(C.byref(self._h), arr, len(self.devices), W, H, self.first_shard, self.total_shards))
            note = self._L.pt_last_error().decode()
            if note.startswith("warning:"):
                import warnings
                warnings.warn(note[len("warning:"):].strip())
        else:
            _check(self._L.pt_create(C.byref(self._h), device, W, H, shard_rank, shard_count))

    def close(self):
        if getattr(self, "_h", None):
            self._L.pt_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # --- SSBO / texture uploads -------------------------------------------------------------
    def set_buffer(self, binding, array):
        a = np.ascontiguousarray(array)
        assert a.dtype in (np.float32, np.int32), "SSBO contents are float32 or int32"
        _check(self._L.pt_set_buffer(self._h, int(binding), a.ctypes.data, a.nbytes))

    def set_texture(self, index, rgba8):
        a = np.ascontiguousarray(rgba8, dtype=np.uint8)
        assert a.ndim == 3 and a.shape[2] == 4
        _check(self._L.pt_set_texture(self._h, int(index), a.shape[1], a.shape[0], a.ctypes.data))

    def load_workload(self, wl):
        for b, arr in wl.buffers.items():
            self.set_buffer(b, arr)
        self.set_texture(0, wl.sky)
        for idx, arr in getattr(wl, "textures", {}).items():
            self.set_texture(idx, arr)

    # --- frame loop ---------------------------------------------------------------------------
    def reset_frame(self):
        _check(self._L.pt_reset_frame(self._h))

    def render(self, frame_count, seed):
        _check(self._L.pt_render(self._h, int(frame_count), int(seed)))

    def render_batch(self, first_frame, seeds):
        s = np.ascontiguousarray(seeds, dtype=np.int32)
        _check(self._L.pt_render_batch(self._h, int(first_frame), int(s.size), s.ctypes.data))

    # overlapped batches: the path pool keeps running from one batch into the next (include/pt_api.h)
    def render_batch_async(self, first_frame, seeds):
        s = np.ascontiguousarray(seeds, dtype=np.int32)
        _check(self._L.pt_render_batch_async(self._h, int(first_frame), int(s.size), s.ctypes.data))

    def next_image(self):
        _check(self._L.pt_next_image(self._h))

    def finish_image(self, age=0):
        _check(self._L.pt_finish_image(self._h, int(age)))

    def image_device(self, age=0):
        p, n = C.c_void_p(), C.c_size_t()
        _check(self._L.pt_image_device(self._h, int(age), C.byref(p), C.byref(n)))
        return p.value, n.value

    def gather_image(self, age=0):
        """Device pointer of the whole W x H RGBA32F image `age` images ago: on a multi-GPU context this is the ONE RCCL gather
        + un-tiling on devices[0] (stream-ordered there, not synchronised)."""
        p = C.c_void_p()
        _check(self._L.pt_gather_image(self._h, int(age), C.byref(p)))
        return p.value

    def synchronize(self):
        _check(self._L.pt_synchronize(self._h))

    def stream_wait(self):
        """wait for what is enqueued on the context's stream(s) (gather, un-tiling, accumulation) without completing batches in flight"""
        _check(self._L.pt_stream_wait(self._h))

    def read_frame(self, out=None):
        if out is None:
            out = np.zeros((self.H, self.W, 4), dtype=np.float32)
        assert out.dtype == np.float32 and out.flags.c_contiguous and out.size == self.W * self.H * 4
        _check(self._L.pt_read_frame(self._h, out.ctypes.data))
        return out

    def write_frame(self, frame):
        """restore a saved FRAME image (running sum + count): the next frames accumulate on top of it (pt_write_frame)"""
        frame = np.ascontiguousarray(frame, dtype=np.float32)
        assert frame.size == self.W * self.H * 4
        _check(self._L.pt_write_frame(self._h, frame.ctypes.data))

    def read_display(self, frame_count, java_bytes=True):
        """The reference's screenshot image: (H, W, 3) uint8, top row first (functions.screenshot, dispatch.java:804-851)."""
        out = np.zeros((self.H, self.W, 3), dtype=np.uint8)
        _check(self._L.pt_read_display(self._h, int(frame_count), 1 if java_bytes else 0, out.ctypes.data))
        return out

    def screenshot(self, path, frame_count, java_bytes=True):
        """functions.screenshot(fileName) (dispatch.java:804-851): the display image as a PNG file, written by the library (pt_save_png)"""
        _check(self._L.pt_save_png(self._h, int(frame_count), 1 if java_bytes else 0, str(path).encode()))

    def frame_device(self):
        p, n = C.c_void_p(), C.c_size_t()
        _check(self._L.pt_frame_device(self._h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def unshard(self, gathered_ptr, full_ptr):
        _check(self._L.pt_unshard(self._h, C.c_void_p(gathered_ptr), C.c_void_p(full_ptr)))

    def set_stream(self, hip_stream):
        _check(self._L.pt_set_stream(self._h, C.c_void_p(hip_stream)))

    def set_option(self, name, value):
        _check(self._L.pt_set_option(self._h, OPTIONS[name], int(value)))

    # --- statistics ---------------------------------------------------------------------------
    def counters(self):
        out = np.zeros(len(COUNTERS), dtype=np.uint64)
        _check(self._L.pt_get_counters(self._h, out.ctypes.data, len(COUNTERS)))
        return dict(zip(COUNTERS, [int(x) for x in out]))

    def reset_counters(self):
        _check(self._L.pt_reset_counters(self._h))

    def set_timing(self, on):
        _check(self._L.pt_set_timing(self._h, 1 if on else 0))

    def kernel_time(self, name):
        n, ms = C.c_int64(), C.c_double()
        _check(self._L.pt_kernel_time(self._h, KERNELS[name], C.byref(n), C.byref(ms)))
        return n.value, ms.value

    def kernel_time_median(self, name):
        ms = C.c_double()
        _check(self._L.pt_kernel_time_median(self._h, KERNELS[name], C.byref(ms)))
        return ms.value

    # --- parity probes --------------------------------------------------------------------------
    def debug_math(self, fn, x, y=None):
        # rng_state / rng_result / rng_random: one NextRandom / random() call of frag.glsl:686-694, the uint32 state travels as float bits
        names = {"sin": 0, "cos": 1, "log": 2, "exp": 3, "atan2": 4, "asin": 5, "rng_state": 6, "rng_result": 7, "rng_random": 8, "unorm8": 9}
        x = np.ascontiguousarray(x, dtype=np.float32)
        out = np.empty_like(x)
        yp = None if y is None else np.ascontiguousarray(y, dtype=np.float32).ctypes.data
        _check(self._L.pt_debug_math(self._h, names[fn], x.ctypes.data, yp, out.ctypes.data, x.size))
        return out

    def debug_intersect(self, o, d):
        o = np.ascontiguousarray(o, dtype=np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(d, dtype=np.float32).reshape(-1, 3)
        out = np.empty((o.shape[0], 4), dtype=np.float32)
        _check(self._L.pt_debug_intersect(self._h, o.ctypes.data, d.ctypes.data, out.ctypes.data, o.shape[0]))
        return out[:, :3].copy(), out[:, 3].copy().view(np.int32)
