#!/usr/bin/env python3
"""Kernel time of one rank's shard (tile_first = r, tile_step = N) at full spp: the per-GPU time of an N-GPU run."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ky_amd import api, dist, _abi as A
lib = A.load_kyhip()
scene = api.cornell_box_scene(A.CB_DEFAULT_SCENE, 1024, 768)
p = api.make_params(1024, 768, 1024, tile_w=int(os.environ.get("TILE", "16")), tile_h=int(os.environ.get("TILE", "16")))
full = None
for N in (1, 2, 4, 8):
    times = []
    for r in range(N):
        t = dist.render_shard(scene, p, r, N, 0); t = dist.render_shard(scene, p, r, N, 0)
        torch.cuda.synchronize()
        times.append(lib.kyhip_kernel_ms(0))
    if N == 1: full = times[0]
    print("N=%d: shard kernel ms min %.2f max %.2f  -> efficiency vs N=1: %.3f" % (N, min(times), max(times), full / (N * max(times))))
