"""CPU, world_size 2 over gloo: the multi-GPU decomposition (interleaved tile shards -> one gather -> de-interleave + add).

The shard renderer here is the CPU oracle standing in for the GPU (there is no GPU in this container); what is under
test is ky_amd.dist's host logic: shard ownership, equal-size tile buffers, the single gather, and the de-interleave.
"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as tdist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tiles_from_film(film, params, rank, world, dist):
    """Pack this rank's tiles of a full-size film into the compact [max_tiles, th, tw, 3] layout render_shard() produces."""
    tw, th = params.tile_w, params.tile_h
    out = torch.zeros((dist.shard_tile_count(params, 0, world), th, tw, 3), dtype=torch.float32)
    for k in range(dist.shard_tile_count(params, rank, world)):
        t = params.tile_first + (rank + k * world) * params.tile_step
        x0, y0 = dist.tile_origin(params, t)
        w, h = min(tw, params.width - x0), min(th, params.height - y0)
        out[k, :h, :w] = torch.from_numpy(film[y0:y0 + h, x0:x0 + w].copy())
    return out


def _worker(rank, world, port, W, H, spp, tile, q, tile_first=0, tile_step=1):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    tdist.init_process_group("gloo", rank=rank, world_size=world)
    from ky_amd import _abi as A, api, dist
    from oracle import kyoracle as O
    scene = api.cornell_box_scene(A.CB_DEFAULT_SCENE, W, H)
    params = api.make_params(W, H, spp, tile_w=tile[0], tile_h=tile[1], tile_first=tile_first, tile_step=tile_step)
    mine = O.render(scene, dist.shard_params(params, rank, world), threads=2)
    tiles = _tiles_from_film(mine, params, rank, world, dist)
    gathered = dist.gather_tiles(tiles, rank, world)
    if rank == 0:
        film = torch.zeros((H, W, 3), dtype=torch.float32)
        dist.add_tiles_to_film(film, gathered, params, world)
        full = O.render(scene, params, threads=2)
        q.put((bool(np.array_equal(film.numpy(), full)), float(np.abs(film.numpy() - full).max())))
    tdist.barrier()
    tdist.destroy_process_group()


@pytest.mark.parametrize("world,size,tile,first,step", [(2, (40, 24), (16, 8), 0, 1), (2, (33, 17), (8, 8), 0, 1), (3, (40, 24), (16, 16), 0, 1),
                                                        (2, (40, 24), (8, 8), 1, 3)])   # the last: the frame is itself a shard (every third tile from 1)
def test_sharded_render_gather_deinterleave(world, size, tile, first, step):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, size[0], size[1], 2, tile, q, first, step)) for r in range(world)]
    for p in procs:
        p.start()
    same, maxdiff = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert same, maxdiff


def test_shard_ownership_is_a_partition():
    sys.path.insert(0, ROOT)
    from ky_amd import api, dist
    p = api.make_params(100, 70, 4, tile_w=32, tile_h=32)   # 4 x 3 tiles
    for world in (1, 2, 3, 5, 8, 16):
        counts = [dist.shard_tile_count(p, r, world) for r in range(world)]
        assert sum(counts) == dist.tiles_total(p) == 12
        assert max(counts) == counts[0] and max(counts) - min(counts) <= 1
        sp = dist.shard_params(p, world - 1, world)
        assert (sp.tile_first, sp.tile_step) == (world - 1, world) and (p.tile_first, p.tile_step) == (0, 1)
    # a frame that is itself a shard: the ranks split ITS tiles (tile_first + r * tile_step, tile_step * world)
    q = api.make_params(100, 70, 4, tile_w=16, tile_h=16, tile_first=2, tile_step=3)   # 7 x 5 = 35 tiles, 11 of them in the frame
    for world in (1, 2, 4):
        counts = [dist.shard_tile_count(q, r, world) for r in range(world)]
        assert sum(counts) == 11 and max(counts) == counts[0]
        sp = dist.shard_params(q, world - 1, world)
        assert (sp.tile_first, sp.tile_step) == (2 + (world - 1) * 3, 3 * world)
