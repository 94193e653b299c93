"""Multi-GPU image sharding, process-per-GPU form: one RCCL collective per sample batch.

(The single-process form lives inside the library: pt_create_multi / pt_gather_image, csrc/hip/pt_multi.hpp.)

Every pixel-frame lane is independent (shared read-only scene, RNG keyed on the GLOBAL pixel
index, /root/reference/src/shaders/frag.glsl:886,896), so the image is cut into 32x8 tiles dealt
round-robin to the ranks, the scene is replicated, and the only exchange is ONE gather of the
packed RGBA32F accumulators at the end of a batch.  torch.distributed is the
plumbing: backend "nccl" is RCCL over xGMI on the GPU box, "gloo" in the CPU tests.
"""
import numpy as np
import torch
import torch.distributed as dist


class _DevArray:
    """zero-copy view of a device pointer for torch.as_tensor (CUDA array interface v2)"""

    def __init__(self, ptr, shape):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": "<f4", "data": (int(ptr), False), "version": 2}


def frame_tensor(renderer, device, age=0):
    """The shard's packed accumulator (n_slots, 4) as a torch tensor aliasing the library's device memory
    (age 1: the previous image of the two that alternate under pt_next_image)."""
    ptr, n = renderer.image_device(age)
    return torch.as_tensor(_DevArray(ptr, (n, 4)), device=device)


def all_maps(W, H, world, shard_map_fn):
    """(world * n_slots,) global pixel index of every packed slot of every rank, -1 = padding"""
    return np.concatenate([shard_map_fn(W, H, r, world) for r in range(world)])


class Unsharder:
    """Un-tiling of the gathered accumulators: packed slot -> global pixel.  On the GPU this is the library's own kernel
    (pt_unshard / k_unshard on the renderer's stream, the same one the multi-GPU context runs after its ncclGather); the torch
    index form is what the CPU (gloo) tests exercise."""

    def __init__(self, W, H, world, shard_map_fn, device, renderer=None, ranks=None, sync_before_unshard=False):
        """world: tile shards of the image; ranks: processes that gather (default: one shard per process); sync_before_unshard: the
        renderer's streams are not torch's (a multi-stream context), so the collective's result is waited for before the un-tiling kernel"""
        maps = torch.from_numpy(all_maps(W, H, world, shard_map_fn).astype(np.int64))
        self.W, self.H, self.shards = W, H, world
        self.world = world if ranks is None else ranks
        self.sync = sync_before_unshard
        self.renderer = renderer if torch.device(device).type == "cuda" else None
        self.src = torch.nonzero(maps >= 0).squeeze(1).to(device)          # packed slots that carry a pixel
        self.dst = maps[maps >= 0].to(device)                               # their global pixel indices
        self.identity = world == 1 and bool((self.src.cpu() == self.dst.cpu()).all())      # one shard: the packed accumulator is the image
        self._full = None

    to_device = None        # rehearsals with a CPU collective: the gathered blocks go to this device for the library's un-tiling kernel

    def __call__(self, gathered):
        if self.to_device is not None and not gathered.is_cuda:
            gathered = gathered.to(self.to_device)
        if self.identity:
            return gathered[: self.W * self.H].reshape(self.H, self.W, 4)
        if self.renderer is not None and gathered.is_cuda:
            if self._full is None:
                self._full = torch.empty((self.H * self.W, 4), dtype=gathered.dtype, device=gathered.device)
            if self.sync:
                torch.cuda.current_stream(gathered.device).synchronize()
            self.renderer.unshard(gathered.data_ptr(), self._full.data_ptr())      # stream-ordered on the renderer's stream
            return self._full.reshape(self.H, self.W, 4)
        full = torch.empty((self.H * self.W, 4), dtype=gathered.dtype, device=gathered.device)
        full.index_copy_(0, self.dst, gathered.index_select(0, self.src))
        return full.reshape(self.H, self.W, 4)


def gather_frame(packed, unsharder, dst=0, group=None, force_collective=False):
    """ONE collective: gather every rank's packed accumulator (n_slots,4) on `dst`, then un-tile.

    Returns the full (H, W, 4) image on rank `dst`, None elsewhere.  force_collective: issue the gather even in a world of one
    (bench.py --dist: the RCCL call path of the process-per-GPU form, exercised on a one-GPU box)."""
    world = unsharder.world
    if world == 1 and not (force_collective and dist.is_initialized()):
        return unsharder(packed)
    rank = dist.get_rank(group)
    if rank == dst:
        gathered = torch.empty((world,) + tuple(packed.shape), dtype=packed.dtype, device=packed.device)
        dist.gather(packed, list(gathered.unbind(0)), dst=dst, group=group)
        return unsharder(gathered.reshape(-1, 4))
    dist.gather(packed, None, dst=dst, group=group)
    return None


class StepPipeline:
    """bench.py's schedule for consecutive images: every step starts a new image (pt_next_image) and submits its batches
    asynchronously; the image is completed and gathered LAG steps later, when its last paths have long retired, so neither the
    path pool nor the collective ever waits for a straggler.  `drain()` completes and gathers what is still in flight."""

    def __init__(self, renderer, unsharder, device, lag=2, dst=0, group=None, tensor_of=None, collect=None, force_collective=False):
        assert 0 <= lag <= 3, "the library keeps a ring of four FRAME images"
        self.r, self.unsharder, self.device, self.lag, self.dst, self.group = renderer, unsharder, device, lag, dst, group
        self.tensor_of = tensor_of or (lambda r, age: frame_tensor(r, device, age))
        self.collect = collect          # collect(age) -> image: a multi-GPU context gathers inside the library (pt_gather_image)
        self.force_collective = force_collective
        self.in_flight = 0

    def _collect(self, age):
        if self.collect is not None:
            return self.collect(age)
        self.r.finish_image(age)
        return gather_frame(self.tensor_of(self.r, age), self.unsharder, dst=self.dst, group=self.group, force_collective=self.force_collective)

    def step(self, submit):
        """submit(): the step's pt_render_batch_async calls.  Returns the gathered image of the step LAG steps ago (rank dst), else None."""
        self.r.next_image()
        submit()
        if self.in_flight == self.lag:
            return self._collect(self.lag)
        self.in_flight += 1
        return None

    def drain(self):
        """-> the gathered images still in flight, oldest first"""
        out = []
        while self.in_flight > 0:
            self.in_flight -= 1
            out.append(self._collect(self.in_flight))
        return out
