"""CPU, world_size 2 (gloo): the N > 1 host path — tile sharding, ONE gather of the packed accumulators, un-tiling.

The per-rank render is stood in for by the oracle (allowed in tests only): each rank fills its packed,
shard-local accumulator exactly the way the HIP renderer lays it out (pt_shard_map order), then runs
the same shard.gather_frame() that bench.py runs over RCCL.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, W, H, out_path):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import ptimport
    pt = ptimport.load()
    from pathtracer_0_amd import renderer, shard
    import oracle
    wl = pt.scenes.build("C2", W, H)
    seeds = [pt.scenes.frame_seed(f) for f in (1, 2)]
    full, _ = oracle.render_frames(oracle.Scene.from_workload(wl), W, H, 1, 2, seeds, nthreads=1)
    m = renderer.shard_map(W, H, rank, world)
    packed = np.zeros((len(m), 4), np.float32)
    packed[m >= 0] = full.reshape(-1, 4)[m[m >= 0]]            # what this rank's renderer would hold
    un = shard.Unsharder(W, H, world, renderer.shard_map, torch.device("cpu"))
    got = shard.gather_frame(torch.from_numpy(packed), un, dst=0)
    if rank == 0:
        np.save(out_path, np.stack([got.numpy(), full]))
    else:
        assert got is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("W,H", [(96, 40), (100, 37)])
def test_two_rank_gather_reassembles_the_frame(pt, tmp_path, W, H):
    from pathtracer_0_amd import build
    build.build_hip()
    out = str(tmp_path / "res.npy")
    mp.spawn(_worker, args=(2, _free_port(), W, H, out), nprocs=2, join=True)
    got, full = np.load(out)
    assert np.array_equal(got, full)


def _worker_two_shards_per_rank(rank, world, port, W, H, out_path):
    """bench.py's process-per-GPU form as the driver's scaling runs start it: every rank holds K = 2 streams = tile shards rank*K, rank*K+1 of world*K and gathers
    ONE block of K packed accumulators (pt_gather_image of a part group); rank 0 un-tiles world*K shards (Unsharder(..., world*K, ranks=world))"""
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import ptimport
    pt = ptimport.load()
    from pathtracer_0_amd import renderer, shard
    K = 2
    rng = np.random.default_rng(5)                                # the same synthetic image on every rank: every pixel distinct, so a misplaced slot shows
    full = rng.random((H * W, 4), dtype=np.float32)
    ns = renderer.shard_slots(W, H, world * K)
    block = np.zeros((K * ns, 4), np.float32)
    for k in range(K):
        m = renderer.shard_map(W, H, rank * K + k, world * K)
        assert len(m) == ns                                     # every shard padded to the same size: the collective's blocks are equal
        block[k * ns:(k + 1) * ns][m >= 0] = full[m[m >= 0]]
    un = shard.Unsharder(W, H, world * K, renderer.shard_map, torch.device("cpu"), ranks=world)
    got = shard.gather_frame(torch.from_numpy(block), un, dst=0)
    if rank == 0:
        np.save(out_path, np.stack([got.numpy().reshape(-1, 4), full]))
    else:
        assert got is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,W,H", [(8, 100, 37), (8, 1920, 1080), (2, 96, 40)])
def test_ranks_with_two_shards_each_reassemble_the_frame(pt, tmp_path, world, W, H):
    """world 8 x 2 streams = the 16 tile shards of the 8-GPU run (no render: the gather layout and the un-tiling, at an odd size with fewer
    tiles than two per shard and at the headline size)"""
    from pathtracer_0_amd import build
    build.build_hip()
    out = str(tmp_path / "res.npy")
    mp.spawn(_worker_two_shards_per_rank, args=(world, _free_port(), W, H, out), nprocs=world, join=True)
    got, full = np.load(out)
    assert np.array_equal(got, full)


class _OracleRenderer:
    """Stands in for one rank's renderer in the pipeline test: same call sequence as renderer.Renderer (next_image,
    render_batch_async, finish_image) over a ring of four shard-local accumulators, the frames coming from the oracle."""

    def __init__(self, oracle, scene, W, H, shard_map):
        self.oracle, self.scene, self.W, self.H, self.map = oracle, scene, W, H, shard_map
        self.ring = [np.zeros((len(shard_map), 4), np.float32) for _ in range(4)]
        self.cur, self.pending, self.log = 0, [], []

    def next_image(self):
        self.cur = (self.cur + 1) % 4
        assert all(img != self.cur for img, _, _ in self.pending), "an image was taken over while batches were still on their way into it"
        self.ring[self.cur][:] = 0
        self.log.append("next")

    def render_batch_async(self, first, seeds):
        self.pending.append((self.cur, first, list(seeds)))       # nothing lands before finish_image: the worst case the schedule must survive
        self.log.append("submit")

    def finish_image(self, age):
        img = (self.cur - age) % 4
        for (i, first, seeds) in [p for p in self.pending if p[0] == img]:
            full = np.zeros((self.H, self.W, 4), np.float32)
            m = self.map
            full.reshape(-1, 4)[m[m >= 0]] = self.ring[img][m >= 0]
            self.oracle.render_frames(self.scene, self.W, self.H, first, len(seeds), seeds, frame=full, nthreads=1)
            self.ring[img][m >= 0] = full.reshape(-1, 4)[m[m >= 0]]
        self.pending = [p for p in self.pending if p[0] != img]
        self.log.append(f"finish{age}")

    def image(self, age):
        return self.ring[(self.cur - age) % 4]


def _pipeline_worker(rank, world, port, W, H, steps, lag, out_path):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import ptimport
    pt = ptimport.load()
    from pathtracer_0_amd import renderer, shard
    import oracle
    wl = pt.scenes.build("C2", W, H)
    sc = oracle.Scene.from_workload(wl)
    r = _OracleRenderer(oracle, sc, W, H, renderer.shard_map(W, H, rank, world))
    un = shard.Unsharder(W, H, world, renderer.shard_map, torch.device("cpu"))
    pipe = shard.StepPipeline(r, un, torch.device("cpu"), lag=lag, tensor_of=lambda rr, age: torch.from_numpy(rr.image(age).copy()))
    seeds = [[(37 * k + 11 * f) % 10000 for f in (1, 2)] for k in range(steps)]
    got = []
    for k in range(steps):
        def submit(k=k):
            r.render_batch_async(1, seeds[k][:1]); r.render_batch_async(2, seeds[k][1:])
        out = pipe.step(submit)
        if out is not None:
            got.append(out)
    got += [g for g in pipe.drain() if g is not None]
    if rank == 0:
        ref = [oracle.render_frames(sc, W, H, 1, 2, seeds[k], nthreads=1)[0] for k in range(steps)]
        np.save(out_path, np.stack([np.stack([g.numpy() for g in got]), np.stack(ref)]))
    else:
        assert got == []
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("steps,lag", [(5, 2), (1, 2), (3, 0), (4, 3)])
def test_two_rank_step_pipeline(pt, tmp_path, steps, lag):
    """bench.py's overlapped schedule (shard.StepPipeline) over gloo: every step's image arrives exactly once, in order, complete"""
    out = str(tmp_path / "pipe.npy")
    mp.spawn(_pipeline_worker, args=(2, _free_port(), 64, 40, steps, lag, out), nprocs=2, join=True)
    got, ref = np.load(out)
    assert got.shape[0] == steps
    assert np.array_equal(got, ref)
