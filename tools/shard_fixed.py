"""Fixed cost of a launch: kernel ms of a 1/8 shard and of the full frame of configs[1]'s scene at very small spp (start-up + one short item per wave)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ky_amd import api, dist, _abi as A
lib = A.load_kyhip()
scene = api.cornell_box_scene(A.CB_DEFAULT_SCENE, 1024, 768)
for spp in (1, 4, 8, 16, 32, 64, 128):
    p = api.make_params(1024, 768, spp)
    out = []
    for N in (8, 1):
        best = 1e9
        for _ in range(5):
            dist.render_shard(scene, p, 0, N, 0); torch.cuda.synchronize()
            best = min(best, lib.kyhip_kernel_ms(0))
        out.append(best)
    print("spp %4d  1/8 shard %.3f ms   full frame %.3f ms   (full / 8 = %.3f)" % (spp, out[0], out[1], out[1] / 8))
