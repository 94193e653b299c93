#!/usr/bin/env python3
"""Which libamdhip64 does a process map?  hip_runtime_probe.py {torch-first|lib-first|lib-only}
Prints every mapped libamdhip64 / libhsa-runtime64 / librccl path and what hipRuntimeGetVersion says through libpt_hip.so's own binding."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
mode = sys.argv[1] if len(sys.argv) > 1 else "torch-first"


def mapped():
    out = set()
    for line in open("/proc/self/maps"):
        p = line.split()[-1]
        if any(k in p for k in ("libamdhip64", "libhsa-runtime64", "librccl", "libpt_hip")):
            out.add(os.path.realpath(p))
    return sorted(out)


def load_lib():
    L = C.CDLL(os.path.join(ROOT, "pathtracer-0_amd", "libpt_hip.so"))
    h = C.c_void_p()
    L.pt_last_error.restype = C.c_char_p
    rc = L.pt_create(C.byref(h), 0, 64, 64, 0, 1)
    print(mode, "pt_create rc", rc, L.pt_last_error().decode() if rc else "")
    return L, h


if mode == "torch-first":
    import torch
    print(mode, "torch.cuda.is_available", torch.cuda.is_available(), "torch.version.hip", torch.version.hip)
    if torch.cuda.is_available():
        torch.zeros(1, device="cuda")
    L, h = load_lib()
elif mode == "lib-first":
    L, h = load_lib()
    import torch
    try:
        print(mode, "torch.cuda.is_available", torch.cuda.is_available())
        torch.zeros(1, device="cuda")
        print(mode, "torch tensor on the GPU: ok")
    except Exception as e:
        print(mode, "torch failed:", type(e).__name__, e)
else:
    L, h = load_lib()
for p in mapped():
    print(mode, "mapped", p)
