"""Build recipes for the native libraries (in-tree, so the .so files travel with the snapshot).

  libpt_host.so  g++    host-side scene producers (csrc/host/scene_host.cpp)
  libpt_hip.so   hipcc  gfx950 render path + C ABI (csrc/hip/pt_hip.hip), GPU BVH builder (csrc/hip/pt_bvh.hip)
"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
# numeric contract: no implicit FMA contraction, IEEE divide/sqrt
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt", "-fPIC", "-shared",
             "-Wall", "-Wno-unused-value"]
HOST_FLAGS = ["-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wall", "-Wextra"]


def _stale(out, srcs):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(s) > t for s in srcs)


def build_host(force=False):
    src = os.path.join(HERE, "csrc", "host", "scene_host.cpp")
    hdr = os.path.join(HERE, "..", "include", "pt_scene.h")
    out = os.path.join(HERE, "libpt_host.so")
    if force or _stale(out, [src, hdr]):
        subprocess.check_call(["g++"] + HOST_FLAGS + ["-o", out, src])
    return out


def build_hip(force=False, extra=()):
    d = os.path.join(HERE, "csrc", "hip")
    srcs = [os.path.join(d, f) for f in ("pt_hip.hip", "pt_bvh.hip", "pt_device.hpp", "pt_math.hpp")] + [os.path.join(HERE, "..", "include", "pt_api.h")]
    out = os.path.join(HERE, "libpt_hip.so")
    if force or _stale(out, srcs):
        hipcc = "/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else "hipcc"
        subprocess.check_call([hipcc] + HIP_FLAGS + list(extra) + ["-o", out, srcs[0], srcs[1]])
    return out


def kernel_source_hash():
    """sha256 (16 hex digits) over the device sources, in name order: committed counter summaries (profiles/pmc_*.json) carry it, and
    bench.py flags them stale when the kernels have changed underneath"""
    import hashlib
    d = os.path.join(HERE, "csrc", "hip")
    h = hashlib.sha256()
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".hpp", ".s", ".inc")):
            h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def build_all(force=False):
    return build_host(force), build_hip(force)


if __name__ == "__main__":
    print(build_all(force=True))
