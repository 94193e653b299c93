"""GPU: the multi-stream context behind the C ABI (pt_create_multi / pt_create_multi_part / pt_gather_image, csrc/hip/pt_multi.hpp) and
bench.py's launch forms.

A one-GPU box cannot run two RCCL ranks (RCCL refuses two ranks on one device), so:
  * devices=[0, 0] / [0, 0, 0] ...: several independent streams on one GPU (the production form: two streams per GPU) — one host thread
    per stream, replicated scene, image ring, shard maps, padding, device-copy gather, un-tiling kernel;
  * the same with PT_MULTI_FORCE_RCCL=1: the RCCL call path of the several-GPU form on one device — dlopen(librccl), ncclCommInitAll,
    the streams' accumulators staged side by side, ONE ncclGather of the block in a group call, un-tiling;
  * pt_create_multi_part: two groups holding shards 0-1 and 2-3 of 4 (one process per GPU, two streams each) + pt_unshard.
Everything must be bit-identical to the plain one-stream context (K10, frag.glsl:886,896: the RNG is keyed on the global pixel).
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


@pytest.mark.parametrize("devices,virtual,W,H", [([0, 0, 0, 0], 2, 100, 37), ([0, 0, 0], 3, 96, 54), ([0, 0, 0, 0, 0, 0, 0, 0], 4, 160, 90), ([0, 0, 0, 0, 0, 0], 2, 70, 41),
                                                 # the production form of an 8-GPU node: 8 devices x 2 streams = 16 shards — at a size with fewer tiles (20) than two per shard,
                                                 # odd edges, and at the headline size
                                                 ([0] * 16, 8, 100, 37), ([0] * 16, 8, 1920, 1080)])
def test_several_device_gather_on_virtual_devices(pt, renderer_mod, devices, virtual, W, H, monkeypatch):
    """The several-DEVICE code of the gather (pt_multi.hpp: one staging block per device — none for a device with one stream —, root and
    non-root arguments of the collective, block offsets, un-tiling of the gathered buffer) on a one-GPU box: PT_MULTI_VIRTUAL_DEVICES=k
    splits the streams of GPU 0 into k devices and replaces only the RCCL calls, by device copies to the offsets ncclGather writes.
    {0,0|0,0} is the production shape of a 2-GPU node (two streams per GPU), {0|0|0} three GPUs with one stream each."""
    monkeypatch.setenv("PT_MULTI_VIRTUAL_DEVICES", str(virtual))
    wl = pt.scenes.build("C3", W, H)
    seeds = _seeds(pt, 3 if W * H < 1000000 else 1)
    rm = renderer_mod.Renderer(W, H, devices=devices)
    monkeypatch.delenv("PT_MULTI_VIRTUAL_DEVICES")
    rm.load_workload(wl); rm.reset_frame(); rm.render_batch(1, seeds)
    got = rm.read_frame().copy()
    disp = rm.read_display(len(seeds))
    rm.next_image()
    for k in range(0, len(seeds), 2):
        rm.render_batch_async(1 + k, seeds[k:k + 2])
    again = rm.read_frame().copy()                              # a second image through the same staging blocks
    rm.close()
    r1 = renderer_mod.Renderer(W, H)
    r1.load_workload(wl); r1.reset_frame(); r1.render_batch(1, seeds)
    ref = r1.read_frame(); ref_disp = r1.read_display(len(seeds)); r1.close()
    assert np.array_equal(got, ref, equal_nan=True) and np.array_equal(again, ref, equal_nan=True)
    assert np.array_equal(disp, ref_disp)


def test_virtual_devices_argument_checks(renderer_mod, monkeypatch):
    monkeypatch.setenv("PT_MULTI_VIRTUAL_DEVICES", "2")
    with pytest.raises(renderer_mod.PtError):
        renderer_mod.Renderer(64, 64, devices=[0, 0, 0])         # 3 streams do not split into 2 devices


@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0]])
def test_multi_context_through_rccl(pt, renderer_mod, devices, monkeypatch):
    """the gather of the several-GPU form (staging block per device + ONE ncclGather + un-tiling), forced onto the one device of this box"""
    monkeypatch.setenv("PT_MULTI_FORCE_RCCL", "1")
    W, H = 100, 37
    wl = pt.scenes.build("C2", W, H)
    seeds = _seeds(pt, 2)
    rm = renderer_mod.Renderer(W, H, devices=devices)
    rm.load_workload(wl); rm.reset_frame(); rm.render_batch(1, seeds)
    got = rm.read_frame().copy()
    rm.next_image(); rm.render_batch_async(1, seeds[:1]); rm.render_batch_async(2, seeds[1:])
    again = rm.read_frame().copy()                              # a second gather through the same communicator
    rm.close()
    monkeypatch.delenv("PT_MULTI_FORCE_RCCL")
    r1 = renderer_mod.Renderer(W, H)
    r1.load_workload(wl); r1.reset_frame(); r1.render_batch(1, seeds)
    ref = r1.read_frame(); r1.close()
    assert np.array_equal(got, ref, equal_nan=True) and np.array_equal(again, ref, equal_nan=True)


@pytest.mark.parametrize("P,W,H", [(2, 160, 90), (8, 100, 37), (8, 640, 360)])
def test_part_groups_and_unshard(pt, renderer_mod, P, W, H):
    """one process per GPU with two streams each, here all "processes" on GPU 0: groups of shards 2p, 2p+1 of 2P; each packs its block
    (pt_gather_image), the blocks side by side are what the processes' collective delivers, pt_unshard rebuilds the image.  P = 8 is the
    driver's 8-GPU scaling run: 16 shards (at 100x37 there are 20 tiles for them: some shards hold one tile, the blocks are mostly padding)"""
    import torch
    from pathtracer_0_amd import shard
    wl = pt.scenes.build("C3", W, H)
    seeds = _seeds(pt, 3)
    ns = renderer_mod.shard_slots(W, H, 2 * P)
    parts, blocks = [], []
    acc = np.zeros((H, W, 4), np.float32)
    for p in range(P):
        g = renderer_mod.Renderer(W, H, devices=[0, 0], first_shard=2 * p, total_shards=2 * P)
        g.load_workload(wl); g.reset_frame(); g.render_batch(1, seeds)
        blk = torch.as_tensor(shard._DevArray(g.gather_image(0), (2 * ns, 4)), device="cuda:0")
        g.stream_wait()
        blocks.append(blk.clone()); parts.append(g)
        g.read_frame(acc)                                        # a part group writes only its own pixels
    gathered = torch.cat(blocks).contiguous()
    full = torch.empty((H * W, 4), dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    parts[0].unshard(gathered.data_ptr(), full.data_ptr())
    parts[0].stream_wait()
    got = full.cpu().numpy().reshape(H, W, 4)
    for g in parts:
        g.close()
    r1 = renderer_mod.Renderer(W, H)
    r1.load_workload(wl); r1.reset_frame(); r1.render_batch(1, seeds)
    ref = r1.read_frame(); r1.close()
    assert np.array_equal(got, ref, equal_nan=True) and np.array_equal(acc, ref, equal_nan=True)


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


def test_multi_stream_scheduler_paths(pt, oracle, renderer_mod):
    """the frame-stream scheduler under a two-stream context: overlapped images, a Parameters upload and a camera move between asynchronous
    batches (each ends the running streams of BOTH shards first), synchronous calls in between — against the oracle, frame by frame"""
    W, H = 96, 54
    wl = pt.scenes.build("C3", W, H)
    wl2 = wl.with_params(SAMPLE_RES=4, MAX_BOUNCES=2)
    moved = wl.buffers[0].copy(); moved[0] += 0.2
    r = renderer_mod.Renderer(W, H, devices=[0, 0])
    r.set_option("path_slots", 1024)                           # per stream: deep backlogs
    r.load_workload(wl); r.reset_frame()
    r.render_batch_async(1, [11, 22, 33])
    r.set_buffer(4, wl2.buffers[4]); r.render_batch_async(4, [44])
    r.set_buffer(4, wl.buffers[4]); r.set_buffer(0, moved); r.render(5, 55)
    r.render_batch_async(6, [66, 77])
    got = r.read_frame().copy()
    cnt_ok = r.counters()["iterations"] > 0
    r.close()
    sc = oracle.Scene.from_workload
    ref = np.zeros((H, W, 4), np.float32)
    oracle.render_frames(sc(wl), W, H, 1, 3, [11, 22, 33], frame=ref, nthreads=8)
    oracle.render_frames(sc(wl2), W, H, 4, 1, [44], frame=ref, nthreads=8)
    b = dict(wl.buffers); b[0] = moved
    wl3 = pt.scenes.Workload(wl.name, W, H, b, wl.sky, wl.sample_res, wl.max_bounces, wl.info)
    oracle.render_frames(sc(wl3), W, H, 5, 1, [55], frame=ref, nthreads=8)
    oracle.render_frames(sc(wl3), W, H, 6, 2, [66, 77], frame=ref, nthreads=8)
    assert cnt_ok and np.array_equal(got, ref, equal_nan=True)


def test_multi_context_errors(pt, renderer_mod):
    with pytest.raises(renderer_mod.PtError) as e:
        renderer_mod.Renderer(64, 48, devices=[0, 99])       # no such device
    assert e.value.code == -2
    with pytest.raises(renderer_mod.PtError) as e:
        renderer_mod.Renderer(64, 48, devices=[0, 0], first_shard=3, total_shards=4)       # shards 3 and 4 of 4
    assert e.value.code == -1
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


def test_bench_default_is_two_streams_on_one_gpu():
    """`python bench.py` (the driver's N = 1 command): one GPU, two independent wavefront streams behind one context, checked against the oracle"""
    d = _bench("--gpus", "1")
    assert d["n_gpus"] == 1 and "2 independent wavefront stream(s) per GPU" in d["config"]["multi_gpu"]
    assert d["parity"]["bit_identical"] is True and d["cpu_baseline"]["value"] > 0
    r = d["roofline"]
    assert r["bound"] == "vmem_divergent" and r["streams_per_gpu"] == 2 and 0 < r["chip_valu_issue"]["frac"] <= 1 and 0 < r["hbm"]["frac"] <= 1 and 0 < r["valu_issue"]["frac"] <= 1
    if r["counters_stale"]:      # the committed counter summary belongs to other kernel sources (a tree between two measurement passes): the headline fractions are withheld
        assert r["frac"] is None and r["hbm_frac"] is None and d["hbm_frac"] is None and "COUNTERS STALE" in r["note"] and r["valu_issue"]["provisional"] is True
    else:
        assert 0 < r["frac"] <= 1 and r["frac"] == r["vmem"]["td_busy"]["frac"] and r["hbm_frac"] == r["hbm"]["frac"] == d["hbm_frac"]
    assert "measured_in" in r["alone"] and d["readback"]["ms_per_image"] > 0 and d["cpu_baseline"]["cores"] <= d["cpu_baseline"]["cores_available"]


def test_bench_runs_unaided_with_several_gpus_worth_of_streams():
    """`python bench.py --gpus N` started plainly (no launcher) builds ONE context for all streams; here four streams on GPU 0, gathered through RCCL"""
    d = _bench("--gpus", "1", "--devices", "0,0,0,0", "--no-cpu-baseline", env={"PT_MULTI_FORCE_RCCL": "1"})
    assert d["n_gpus"] == 1 and d["value"] > 0
    d = _bench("--gpus", "1", "--streams", "1")
    assert d["parity"]["bit_identical"] is True and "one stream" in d["config"]["multi_gpu"]


def test_bench_under_torch_distributed_run_nccl():
    """the driver's launch form (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`) with N = 1: init_process_group("nccl"),
    dist.gather on the tensor that aliases the library's accumulator, barrier + max-over-ranks timing"""
    d = _bench("--gpus", "1", launcher=("-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", "29533"))
    assert d["n_gpus"] == 1 and "torch.distributed" in d["config"]["multi_gpu"]
    assert d["parity"]["gathered_image_bit_identical_to_one_gpu_render"] is True


def test_bench_two_ranks_rehearsed_on_one_gpu():
    """every line of the driver's N = 2 run (`torch.distributed.run --nproc-per-node 2 bench.py --gpus 2`: part groups of two streams per rank,
    all-reduced statistics, gather of the packed blocks, pt_unshard of four shards on rank 0, barrier + max-over-ranks timing) except RCCL itself,
    which cannot hold two ranks on one device: the collective runs over gloo on CPU copies, both ranks render on GPU 0"""
    d = _bench("--gpus", "2", "--dist-backend", "gloo", env={"PT_BENCH_ONE_GPU": "1", "HSA_ENABLE_IPC_MODE_LEGACY": "0"},
               launcher=("-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29551"))
    assert d["n_gpus"] == 2 and d["parity"]["gathered_image_bit_identical_to_one_gpu_render"] is True and "rehearsal" in d
    assert "4 shard(s)" in d["config"]["workload"] and d["roofline"]["streams_per_gpu"] == 2


def test_bench_spawns_its_own_ranks():
    """`python bench.py --gpus 2 --spawn`, started plainly: the parent starts torch.distributed.run before touching HIP and relays the line"""
    d = _bench("--gpus", "2", "--spawn", "--dist-backend", "gloo", env={"PT_BENCH_ONE_GPU": "1", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    assert d["n_gpus"] == 2 and d["parity"]["gathered_image_bit_identical_to_one_gpu_render"] is True


def test_part_group_path_of_a_two_rank_run():
    """what rank 0 of `torch.distributed.run --nproc-per-node 2` does, without the second GPU: `--dist` at world 1 is covered above; here the
    packed-block path itself — a part group (shards 0-1 of 4), its block gathered by torch.distributed (nccl, world 1) — in a child process"""
    code = (
        "import os,sys,numpy as np,torch,torch.distributed as dist\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "import ptimport; pt=ptimport.load()\n"
        "from pathtracer_0_amd import renderer, scenes, shard\n"
        "os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29547', RANK='0', WORLD_SIZE='1')\n"
        "dev=torch.device('cuda',0); torch.cuda.set_device(0); dist.init_process_group('nccl', device_id=dev)\n"
        "W,H=128,72; wl=scenes.build('C2',W,H)\n"
        "g=renderer.Renderer(W,H,devices=[0,0],first_shard=0,total_shards=4); g.load_workload(wl); g.reset_frame(); g.render_batch(1,[5,6])\n"
        "ns=renderer.shard_slots(W,H,4)\n"
        "blk=torch.as_tensor(shard._DevArray(g.gather_image(0),(2*ns,4)),device=dev); g.stream_wait()\n"
        "out=torch.empty((1,2*ns,4),device=dev); dist.gather(blk, list(out.unbind(0)), dst=0); torch.cuda.synchronize()\n"
        "r1=renderer.Renderer(W,H); r1.load_workload(wl); r1.reset_frame(); r1.render_batch(1,[5,6]); ref=r1.read_frame().reshape(-1,4)\n"
        "m=np.concatenate([renderer.shard_map(W,H,s,4) for s in (0,1)])\n"
        "got=out[0].cpu().numpy()\n"
        "assert np.array_equal(got[m>=0], ref[m[m>=0]], equal_nan=True)\n"
        "g.close(); r1.close(); dist.destroy_process_group(); print('PART_OK')\n")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert "PART_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


@pytest.mark.gpu
def test_intersect_kernel_under_contention():
    """scripts/contention_check.py in a process of its own: while two more contexts render C3 on the same GPU from threads of their own, the same 20 000
    random rays are traced on the hand-written kernel (block shape of the shared-GPU mode) and every hit record compared with the compiled kernel's.
    ONE functional pass (3 launches) by default: the regression guard for the bug class this once reproduced — a load still in flight when its registers
    are reused (round 3's refill race) — is the static scan (rules S and V of scripts/asm_hazards.py with their known-bad snippets, in the CPU suite), where
    a regression is a test failure and not a wild fetch on a shared GPU box.  PT_CONTENTION_LAUNCHES=150 restores the long form for a developer's own box."""
    launches = os.environ.get("PT_CONTENTION_LAUNCHES", "3")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "contention_check.py"), "C3", "20000", launches], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
    assert "0 launches with differences" in out.stdout
