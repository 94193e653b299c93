import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import ptimport  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # torch is imported BEFORE libpt_hip.so is loaded: PyTorch ships its own libamdhip64.so with the SONAME the library is linked against,
    # and a process must run on one HIP runtime.  Library first = ROCm's runtime initialises the GPU, torch then maps a second runtime
    # that finds no device ("No HIP GPUs are available": the round-2 failure of test_overlapped_batches_equal_synchronous, reproduced in
    # profiles/r03_a_hip_runtime_probe.txt).  renderer.lib() enforces the order and refuses a process with two runtimes mapped; the import
    # here only makes the order explicit for the tests that alias device memory as torch tensors.
    # (The streams of a GPU need distinct hardware queues: a variable the HIP runtime reads once, so it is set before anything uses HIP.)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    try:
        import torch  # noqa: F401
    except ImportError:
        pass


@pytest.fixture(scope="session")
def pt():
    mod = ptimport.load()
    from pathtracer_0_amd import build
    build.build_host()
    return mod


@pytest.fixture(scope="session")
def oracle(pt):
    import oracle as orc
    orc.lib()
    return orc


@pytest.fixture(scope="session")
def renderer_mod(pt):
    from pathtracer_0_amd import renderer
    return renderer


def frames_equal(a, b):
    """bit-exact comparison that treats NaN == NaN"""
    a = np.asarray(a); b = np.asarray(b)
    return a.shape == b.shape and np.array_equal(a.view(np.uint32) if a.dtype == np.float32 else a, b.view(np.uint32) if b.dtype == np.float32 else b) or \
        np.array_equal(a, b, equal_nan=True)


def rmse(a, b):
    """per-pixel RMSE over rgb of FRAME.rgb / FRAME.a"""
    ia = (a[..., :3] / np.maximum(a[..., 3:4], 1e-30)).astype(np.float64)
    ib = (b[..., :3] / np.maximum(b[..., 3:4], 1e-30)).astype(np.float64)
    d = ia - ib
    both = ~np.isfinite(ia) & ~np.isfinite(ib) & ((ia == ib) | (np.isnan(ia) & np.isnan(ib)))
    d[both] = 0.0                                  # the reference's own inf/NaN pixels (log(0), 0*inf: SURVEY.md A5), reproduced on both sides
    d[~np.isfinite(d)] = np.inf                    # ... a NaN on one side only is a difference
    return float(np.sqrt(np.mean(d ** 2)))
