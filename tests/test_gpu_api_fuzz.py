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
