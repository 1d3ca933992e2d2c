"""kernel time of a 1/N shard vs spp: slope = per-sample cost, intercept = per-launch fixed cost of the shard; and the kernel-level
efficiency of N = 4 / 8 at 1024 spp (slowest of the first three shards).  usage: tools/shard_scan.py [tile size, default 16]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ky_amd import api, dist, _abi as A
lib = A.load_kyhip()
scene = api.cornell_box_scene(A.CB_DEFAULT_SCENE, 1024, 768)
tile = int(sys.argv[1]) if len(sys.argv) > 1 else 16
t1024 = {}
for N in (1, 4, 8):
    xs, ys = [], []
    for spp in (256, 512, 1024, 2048):
        p = api.make_params(1024, 768, spp, tile_w=tile, tile_h=tile)
        worst = 0
        for r in range(min(N, 3)):
            best = 1e9
            for _ in range(3):
                dist.render_shard(scene, p, r, N, 0); torch.cuda.synchronize()
                best = min(best, lib.kyhip_kernel_ms(0))
            worst = max(worst, best)
        xs.append(spp); ys.append(worst)
        if spp == 1024: t1024[N] = worst
    b, a = np.polyfit(xs, ys, 1)
    print("N=%d: %s  -> %.5f ms/spp (x%d = %.5f), intercept %.3f ms" % (N, " ".join("%.2f" % y for y in ys), b, N, b * N, a))
print("kernel-level efficiency at 1024 spp: N=4 %.4f  N=8 %.4f   (N=1 %.2f ms)" % (t1024[1] / 4 / t1024[4], t1024[1] / 8 / t1024[8], t1024[1]))
small = api.cornell_box_scene(A.CB_DEFAULT_SCENE, 128, 96)
print("film mean of 128x96 at 1024 / 100 spp (the chunk schedule must not show): %.7f %.7f" % (api.render(small, api.make_params(128, 96, 1024)).mean(), api.render(small, api.make_params(128, 96, 100)).mean()))
