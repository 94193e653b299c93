"""Random sequences of the render path's entry points (scripts/api_fuzz.py) against a model built from the oracle: the frame-stream
scheduler under any interleaving of synchronous and overlapped batches, image-ring rotation, input uploads and pool-size changes."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def api_fuzz(pt, oracle):
    spec = importlib.util.spec_from_file_location("api_fuzz", os.path.join(ROOT, "scripts", "api_fuzz.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.gpu
@pytest.mark.parametrize("seed", list(range(1, 25)))
def test_random_call_sequences(api_fuzz, seed):
    assert api_fuzz.one(seed)


@pytest.mark.gpu
def test_the_checker_notices_a_wrong_frame(api_fuzz, monkeypatch):
    """self-test: with one wrong seed in the MODEL of some overlapped batches the comparison must fail for some sequences"""
    monkeypatch.setenv("API_FUZZ_SELFTEST", "1")
    assert not all(api_fuzz.one(seed) for seed in range(1, 41))


@pytest.mark.gpu
def test_no_write_after_free_when_contexts_come_and_go(tmp_path):
    """The sequences that found the HIP runtime writing into a stream it had just destroyed (profiles/r06_f_runtime_write_after_free.txt): a context created and destroyed
    per sequence, bursts of one-frame asynchronous submissions in between — run in a child process under tools/canary_malloc.cpp, an LD_PRELOAD allocator that fills freed
    blocks, parks them and aborts with the allocating library's name when one is written to (ASan cannot enter a process that uses the HIP runtime here).  Before the stream
    pool of pt_hip.hip it reported `WRITE AFTER FREE of the block ..., 920 bytes, allocated from libamdhip64.so` within the first 50 sequences, every time."""
    import subprocess
    import sys
    lib = str(tmp_path / "canary_malloc.so")
    out = subprocess.run(["g++", "-O1", "-fPIC", "-shared", "-o", lib, os.path.join(ROOT, "tools", "canary_malloc.cpp"), "-ldl", "-lpthread"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    env = dict(os.environ); env["LD_PRELOAD"] = lib
    run = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "api_fuzz.py"), "100001", "160"], capture_output=True, text=True, timeout=600, env=env)
    assert "canary_malloc" not in run.stdout + run.stderr, (run.stdout + run.stderr)[-3000:]
    assert run.returncode == 0 and "160 sequences, 0 mismatches" in run.stdout, (run.stdout + run.stderr)[-3000:]
