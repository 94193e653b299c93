"""GPU: the multi-GPU context behind the C ABI (pt_create_multi / pt_gather_image, csrc/hip/pt_multi.hpp) and bench.py's launch forms.

A one-GPU box cannot run two RCCL ranks (RCCL refuses two ranks on one device), so:
  * devices=[0, 0] / [0, 0, 0]: several shards on one GPU — the whole group machinery (one host thread per shard, replicated
    scene, image ring, shard maps, padding, un-tiling kernel) with the gather by device copies;
  * devices=[0]: a group of one — dlopen(librccl), ncclCommInitAll, ncclGather in a group call, un-tiling: the RCCL call path.
Everything must be bit-identical to the plain one-GPU context (K10, frag.glsl:886,896: the RNG is keyed on the global pixel).
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _seeds(pt, n, first=1):
    return [pt.scenes.frame_seed(f) for f in range(first, first + n)]


@pytest.mark.parametrize("devices,W,H", [([0, 0], 96, 54), ([0, 0, 0], 100, 37), ([0], 96, 54), ([0, 0, 0, 0, 0, 0, 0, 0], 160, 90)])
def test_multi_context_equals_single(pt, renderer_mod, devices, W, H):
    wl = pt.scenes.build("C3", W, H)
    seeds = _seeds(pt, 3)
    r1 = renderer_mod.Renderer(W, H)
    r1.load_workload(wl); r1.reset_frame(); r1.set_option("count_stats", 1); r1.reset_counters()
    r1.render_batch(1, seeds)
    ref = r1.read_frame().copy()
    ref_disp = r1.read_display(3)
    ref_cnt = r1.counters()
    r1.close()
    rm = renderer_mod.Renderer(W, H, devices=devices)
    rm.load_workload(wl); rm.reset_frame(); rm.set_option("count_stats", 1); rm.reset_counters()
    rm.render_batch(1, seeds[:2])
    rm.render(3, seeds[2])                                   # the reference's one-frame call on the group
    got = rm.read_frame()
    assert np.array_equal(got, ref, equal_nan=True)
    assert np.array_equal(rm.read_display(3), ref_disp)
    cnt = rm.counters()
    for k in ("segments", "nodes", "tritests", "hitupd", "samples"):
        assert cnt[k] == ref_cnt[k], k                        # the shards together trace exactly the image's paths
    # a second image after a reset, and a camera move (frame inputs are replicated like the scene)
    rm.reset_frame()
    rm.set_buffer(0, np.array([0.1, 1.0, -2.9], np.float32))
    rm.render_batch(1, seeds[:1])
    moved = rm.read_frame().copy()
    rm.close()
    r1 = renderer_mod.Renderer(W, H)
    r1.load_workload(wl); r1.set_buffer(0, np.array([0.1, 1.0, -2.9], np.float32)); r1.reset_frame(); r1.render_batch(1, seeds[:1])
    assert np.array_equal(moved, r1.read_frame(), equal_nan=True)
    r1.close()


def test_multi_context_overlapped_images(pt, renderer_mod):
    """bench.py's schedule on the group: a new image per step (pt_next_image), asynchronous batches, each image gathered LAG steps later"""
    import torch
    from pathtracer_0_amd import shard
    W, H, LAG, STEPS = 128, 72, 2, 5
    wl = pt.scenes.build("C2", W, H)
    step_seeds = [[(101 * k + 7 * f) % 10000 for f in (1, 2, 3)] for k in range(STEPS)]
    rm = renderer_mod.Renderer(W, H, devices=[0, 0, 0])
    rm.load_workload(wl); rm.reset_frame()

    def collect(age):
        t = torch.as_tensor(shard._DevArray(rm.gather_image(age), (H, W, 4)), device="cuda:0")
        rm.synchronize()
        return t.cpu().numpy().copy()
    pipe = shard.StepPipeline(rm, None, None, lag=LAG, collect=collect)
    got = []
    for k in range(STEPS):
        out = pipe.step(lambda k=k: (rm.render_batch_async(1, step_seeds[k][:2]), rm.render_batch_async(3, step_seeds[k][2:])))
        if out is not None:
            got.append(out)
    got += pipe.drain()
    rm.close()
    assert len(got) == STEPS
    r1 = renderer_mod.Renderer(W, H)
    r1.load_workload(wl)
    for k in range(STEPS):
        r1.reset_frame(); r1.render_batch(1, step_seeds[k])
        assert np.array_equal(got[k], r1.read_frame(), equal_nan=True), f"step {k}"
    r1.close()


def test_multi_context_errors(pt, renderer_mod):
    with pytest.raises(renderer_mod.PtError) as e:
        renderer_mod.Renderer(64, 48, devices=[0, 99])       # no such device
    assert e.value.code == -2
    rm = renderer_mod.Renderer(64, 48, devices=[0, 0])
    with pytest.raises(renderer_mod.PtError):
        rm.render(1, 5)                                       # no scene yet: the shard's error comes back through the group, with its device
    assert "device 0" in str(renderer_mod.lib().pt_last_error())
    with pytest.raises(renderer_mod.PtError):
        rm.frame_device()                                     # per-shard accumulators are not the group's image
    rm.close()


def _bench(*flags, env=None, launcher=()):
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None); e.pop("RANK", None); e.pop("LOCAL_RANK", None)
    e.update(env or {})
    cmd = [sys.executable, *launcher, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "0", "--frames-per-step", "2", "--width", "384", "--height", "216", *flags]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=e)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


def test_bench_runs_unaided_with_several_shards():
    """`python bench.py --gpus 2` started plainly (no launcher): the single-process multi-GPU context; here both shards on GPU 0"""
    d = _bench("--gpus", "2", "--devices", "0,0")
    assert d["parity"]["gathered_image_bit_identical_to_one_gpu_render"] is True
    assert d["value"] > 0 and "rehearsal" in d
    assert d["roofline"]["frac"] is None or d["roofline"]["frac"] <= 1.0


def test_bench_group_of_one_goes_through_rccl():
    d = _bench("--gpus", "1", "--devices", "0")
    assert d["n_gpus"] == 1 and d["parity"]["gathered_image_bit_identical_to_one_gpu_render"] is True and "RCCL" in d["config"]["multi_gpu"]


def test_bench_under_torch_distributed_run_nccl():
    """the driver's launch form (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`) with N = 1: init_process_group("nccl"),
    dist.gather on the tensor that aliases the library's accumulator, barrier + max-over-ranks timing"""
    d = _bench("--gpus", "1", launcher=("-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", "29533"))
    assert d["n_gpus"] == 1 and "torch.distributed" in d["config"]["multi_gpu"]
    assert d["parity"]["gathered_image_bit_identical_to_one_gpu_render"] is True
