"""kernel time of the first 1/8 shard and of the whole frame of configs[1] from 1 to 1024 spp: which part of the shard's time is per launch (start + drain of the
last stage's items), which per stage of the chunk schedule.  usage: tools/low_spp_scan.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ky_amd import api, dist, _abi as A
lib = A.load_kyhip()
scene = api.cornell_box_scene(A.CB_DEFAULT_SCENE, 1024, 768)
for N in (8, 1):
    out = []
    for spp in (1, 2, 4, 8, 16, 24, 32, 48, 64, 96, 128, 192, 256, 384, 448, 472, 512, 1024):
        p = api.make_params(1024, 768, spp, tile_w=16, tile_h=16)
        best = 1e9
        for _ in range(4):
            dist.render_shard(scene, p, 0, N, 0); torch.cuda.synchronize()
            best = min(best, lib.kyhip_kernel_ms(0))
        out.append("%d:%.3f" % (spp, best))
    print("N=%d" % N, " ".join(out))
