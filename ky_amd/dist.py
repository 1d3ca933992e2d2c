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
    """This rank's shard of `params`: tiles rank, rank+world, ... (a copy; the input is not modified)."""
    p = A.RenderParams.from_buffer_copy(params)
    p.tile_first = rank
    p.tile_step = world
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
    total = tiles_total(params)
    return 0 if rank >= total else (total - rank + world - 1) // world


class FrameBuffers:
    """Device buffers of a frame, allocated ONCE and reused for every frame of the same geometry: this rank's compact tile
    buffer [max_tiles_per_rank, tile_h, tile_w, 3] (ranks that own one tile fewer never write the last slot, which stays
    zero) and, on rank 0, the gather target [world, max_tiles_per_rank, tile_h, tile_w, 3] the collective writes into
    directly.  Nothing is allocated, stacked or copied per frame."""

    def __init__(self, params, rank, world, device):
        self.key = (params.width, params.height, params.tile_w, params.tile_h, rank, world, str(device))
        self.max_tiles = shard_tile_count(params, 0, world)
        shape = (self.max_tiles, params.tile_h, params.tile_w, 3)
        self.tiles = torch.zeros(shape, dtype=torch.float32, device=device)
        self.gathered = None
        if rank == 0:
            self.gathered = self.tiles.unsqueeze(0) if world == 1 else torch.zeros((world,) + shape, dtype=torch.float32, device=device)
            self.gather_list = None if world == 1 else list(self.gathered.unbind(0))


_buffers = {}


def frame_buffers(params, rank, world, device):
    key = (params.width, params.height, params.tile_w, params.tile_h, rank, world, str(device))
    fb = _buffers.get(key)
    if fb is None:
        fb = _buffers[key] = FrameBuffers(params, rank, world, device)
    return fb


def render_shard(scene, params, rank, world, device_index=0, out=None):
    """Render this rank's tiles on its GPU into `out` (default: a fresh zeroed tensor) [max_tiles_per_rank, tile_h, tile_w, 3]
    -- equal sizes on every rank keep the gather a single call."""
    lib = A.load_kyhip()
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


def gather_tiles(tiles, rank, world, group=None, out=None, out_list=None):
    """The one collective of a frame: gather every rank's tile buffer to rank 0.  Returns [world, ...] on rank 0.
    With `out` ([world, ...]) / `out_list` (its slices) the collective writes straight into that tensor."""
    if world == 1:
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
    """De-interleave gathered[r, k] (tile r + k*world) and ADD into film [H, W, 3] (rank 0 only).

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
            t = r + k * world
            assert t < total
            x0, y0 = tile_origin(params, t)
            w, h = min(tw, W - x0), min(th, H - y0)
            film[y0:y0 + h, x0:x0 + w] += gathered[r, k, :h, :w]
    return film


def render_distributed(scene, params, rank, world, device_index=0, film=None, group=None):
    """integrator_t::render over `world` GPUs.  Returns the film (a CUDA tensor) on rank 0, None elsewhere.
    Per frame: one render launch per rank, ONE gather, one add kernel on rank 0; the buffers are cached (FrameBuffers)."""
    fb = frame_buffers(params, rank, world, torch.device("cuda", device_index))
    render_shard(scene, params, rank, world, device_index, out=fb.tiles)
    gathered = gather_tiles(fb.tiles, rank, world, group, out=fb.gathered, out_list=getattr(fb, "gather_list", None))
    if rank != 0:
        return None
    if film is None:
        film = torch.zeros((params.height, params.width, 3), dtype=torch.float32, device=fb.tiles.device)
    return add_tiles_to_film(film, gathered, params, world, device_index)
