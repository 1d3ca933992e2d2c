"""Multi-GPU rendering: image tiles sharded over the ranks of one node, one gather of film tiles at the end.

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm).  The path shards
naturally: tile t of the film goes to rank t mod world (interleaved, for load balance),
the scene is replicated (< 20 KB), there is NO communication while rendering, and the only collective is
one gather of the clamped tile buffers to rank 0, which de-interleaves them and ADDS them into the film
(film_t::add_color semantics, ky.cpp:1586).  Random numbers are keyed by global pixel / sample ids, so the
image is bit-identical for every world size.

torch is plumbing here: device buffers, streams and the collective.  Radiance is computed by libkyhip.so.
"""
import ctypes as C

import torch

from . import _abi as A
from . import api


def shard_params(params, rank, world):
    """This rank's shard of the frame `params` describes (a copy; the input is not modified).  `params` may itself be a shard
    (tile_first, tile_step) of a larger job: rank r takes every world-th of ITS tiles, exactly the split
    kyhip_film_add_gathered_device undoes."""
    p = A.RenderParams.from_buffer_copy(params)
    p.tile_first = params.tile_first + rank * params.tile_step
    p.tile_step = params.tile_step * world
    return p


def tiles_total(params):
    tx = (params.width + params.tile_w - 1) // params.tile_w
    ty = (params.height + params.tile_h - 1) // params.tile_h
    return tx * ty


def tile_origin(params, t):
    """Pixel origin of tile number t (numbering of include/kyhip.h: row-major with every tile row rotated by its index)."""
    tx = (params.width + params.tile_w - 1) // params.tile_w
    row = t // tx
    col = (t % tx + row) % tx
    return col * params.tile_w, row * params.tile_h


def shard_tile_count(params, rank, world):
    """Tiles of shard `rank` of `world` of the frame `params` describes (tile_first / tile_step of `params` honoured)."""
    total = tiles_total(params)
    first, step = params.tile_first + rank * params.tile_step, params.tile_step * world
    return 0 if first >= total else (total - first + step - 1) // step


class FrameBuffers:
    """Device buffers of a frame, allocated ONCE and reused for every frame of the same geometry: this rank's compact tile
    buffer [max_tiles_per_rank, tile_h, tile_w, 3] (ranks that own one tile fewer never write the last slot, which stays
    zero) and, on rank 0, the gather target [world, max_tiles_per_rank, tile_h, tile_w, 3] the collective writes into
    directly.  Nothing is allocated, stacked or copied per frame."""

    def __init__(self, params, rank, world, device):
        self.max_tiles = shard_tile_count(params, 0, world)
        self.free = None   # pipelined frames: the event after which the main stream no longer reads these buffers
        shape = (self.max_tiles, params.tile_h, params.tile_w, 3)
        self.tiles = torch.zeros(shape, dtype=torch.float32, device=device)
        self.gathered = None
        if rank == 0:
            self.gathered = self.tiles.unsqueeze(0) if world == 1 else torch.zeros((world,) + shape, dtype=torch.float32, device=device)
            self.gather_list = None if world == 1 else list(self.gathered.unbind(0))


_buffers = {}


def frame_buffers(params, rank, world, device, slot=0):
    key = (params.width, params.height, params.tile_w, params.tile_h, params.tile_first, params.tile_step, rank, world, str(device), slot)
    fb = _buffers.get(key)
    if fb is None:
        fb = _buffers[key] = FrameBuffers(params, rank, world, device)
    return fb


def render_shard(scene, params, rank, world, device_index=0, out=None):
    """Render this rank's tiles on its GPU into `out` (default: a fresh zeroed tensor) [max_tiles_per_rank, tile_h, tile_w, 3]
    -- equal sizes on every rank keep the gather a single call."""
    lib = A.load_kyhip()
    if world > 1:
        # run-time instantiations across ranks: table and own kernel differ in the last bit of a pixel, so every rank of a frame must be on the same one.
        # Mode 1 (blocking; the code cache is shared through flock) guarantees that as long as no rank's compile fails; mode 2 switches when each process's
        # own background compile finishes -- a matter of timing, so it is refused here.
        mode = lib.kyhip_set_jit(-1)
        if mode == 2:
            import os
            if os.environ.get("KYHIP_JIT") == "2":
                raise RuntimeError("kyhip_set_jit(2) (asynchronous run-time instantiations) would let the ranks of one frame render on different kernels; use mode 1 with world > 1")
            # the library's own default (round 6: mode 2 in a single-process job; torchrun's WORLD_SIZE already turns it off): a caller that builds its own
            # group gets the table's kernels on every rank
            lib.kyhip_set_jit(0)
            mode = 0
        if mode == 1 and lib.kyhip_jit_failures() > 0:
            raise RuntimeError("a run-time instantiation failed on this rank (%s): its shards would come from another kernel than the other ranks'" % lib.kyhip_jit_status().decode())
    p = shard_params(params, rank, world)
    dev = torch.device("cuda", device_index)
    tiles = out
    if tiles is None:
        tiles = torch.zeros((shard_tile_count(params, 0, world), params.tile_h, params.tile_w, 3), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    rc = lib.kyhip_render_tiles_device(device_index, api._scene_ptr(scene), C.byref(p), C.c_void_p(tiles.data_ptr()), None, 0,
                                       C.c_void_p(stream))
    api._check(rc, lib)
    return tiles


def gather_tiles(tiles, rank, world, group=None, out=None, out_list=None, always_collective=False):
    """The one collective of a frame: gather every rank's tile buffer to rank 0.  Returns [world, ...] on rank 0.
    With `out` ([world, ...]) / `out_list` (its slices) the collective writes straight into that tensor.
    always_collective: a world of one calls the communicator too instead of returning its own buffer (tests/test_rccl_gpu.py: the only way a
    one-GPU box executes the RCCL call this function makes)."""
    if world == 1 and not always_collective:
        return tiles.unsqueeze(0)
    import torch.distributed as dist
    if rank != 0:
        dist.gather(tiles, None, dst=0, group=group)
        return None
    if out is None:
        out = torch.empty((world,) + tuple(tiles.shape), dtype=tiles.dtype, device=tiles.device)
        out_list = None
    if out_list is None:
        out_list = list(out.unbind(0))
    dist.gather(tiles, out_list, dst=0, group=group)
    return out


def add_tiles_to_film(film, gathered, params, world, device_index=0):
    """De-interleave gathered[r, k] (tile k of shard r: shard_params) and ADD into film [H, W, 3] (rank 0 only).

    CUDA tensors go through kyhip_film_add_gathered_device (one kernel for all shards); CPU tensors (the gloo tests) use
    index arithmetic.
    """
    H, W = params.height, params.width
    tw, th = params.tile_w, params.tile_h
    total = tiles_total(params)
    if film.is_cuda:
        lib = A.load_kyhip()
        stream = torch.cuda.current_stream(film.device).cuda_stream
        assert gathered.is_contiguous() and gathered.shape[0] == world
        assert film.stride(2) == 1 and film.stride(1) == 3   # a film or a cell of a larger film (film_grid_t): rows may be strided
        rank_stride = gathered[0].numel()
        rc = lib.kyhip_film_add_gathered_device(device_index, C.byref(params), world, C.c_void_p(gathered.data_ptr()), rank_stride,
                                                C.c_void_p(film.data_ptr()), film.stride(0) // 3, C.c_void_p(stream))
        api._check(rc, lib)
        return film
    for r in range(world):
        for k in range(shard_tile_count(params, r, world)):
            t = params.tile_first + (r + k * world) * params.tile_step   # tile k of shard r of the frame `params` describes
            assert t < total
            x0, y0 = tile_origin(params, t)
            w, h = min(tw, W - x0), min(th, H - y0)
            film[y0:y0 + h, x0:x0 + w] += gathered[r, k, :h, :w]
    return film


class _Pipeline:
    """Two side streams per device, used alternately by consecutive frames (render_distributed(pipeline=True))."""

    def __init__(self, device):
        self.streams = [torch.cuda.Stream(device), torch.cuda.Stream(device)]
        self.turn = 0


_pipelines = {}


def render_distributed(scene, params, rank, world, device_index=0, film=None, group=None, pipeline=False):
    """integrator_t::render over `world` GPUs.  Returns the film (a CUDA tensor) on rank 0, None elsewhere.
    Per frame: one render launch per rank, ONE gather, one add kernel on rank 0; the buffers are cached (FrameBuffers).

    pipeline=True: consecutive frames render on two alternating side streams (libkyhip keeps its launch state per stream), so the
    next frame's kernel starts on the compute units the current frame's persistent kernel drains from; gather and add stay on the
    caller's stream, ordered behind the frame's render by an event.  Images are the same; only the overlap differs."""
    dev = torch.device("cuda", device_index)
    if not pipeline:
        fb = frame_buffers(params, rank, world, dev)
        render_shard(scene, params, rank, world, device_index, out=fb.tiles)
    else:
        pl = _pipelines.get(device_index)
        if pl is None:
            pl = _pipelines[device_index] = _Pipeline(dev)
        slot = pl.turn
        pl.turn ^= 1
        fb = frame_buffers(params, rank, world, dev, slot=1 + slot)
        main, side = torch.cuda.current_stream(dev), pl.streams[slot]
        if fb.free is not None:
            side.wait_event(fb.free)          # the frame before last (same buffers) has been gathered and added
        else:
            side.wait_stream(main)            # first use: the buffers' zero-fill was enqueued on the caller's stream, which may be backed up
        with torch.cuda.stream(side):
            render_shard(scene, params, rank, world, device_index, out=fb.tiles)
            done = side.record_event()
        main.wait_event(done)
    gathered = gather_tiles(fb.tiles, rank, world, group, out=fb.gathered, out_list=getattr(fb, "gather_list", None))
    if pipeline:
        if rank != 0:
            fb.free = torch.cuda.current_stream(dev).record_event()
    if rank != 0:
        return None
    if film is None:
        film = torch.zeros((params.height, params.width, 3), dtype=torch.float32, device=fb.tiles.device)
    add_tiles_to_film(film, gathered, params, world, device_index)
    if pipeline:
        fb.free = torch.cuda.current_stream(dev).record_event()
    return film
