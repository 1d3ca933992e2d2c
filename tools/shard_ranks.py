import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from ky_amd import api, dist, _abi as A
lib = A.load_kyhip()
scene = api.cornell_box_scene(A.CB_DEFAULT_SCENE, 1024, 768)
for spp in (1024, 4096):
    p = api.make_params(1024, 768, spp)
    ts = []
    for r in range(8):
        best = 1e9
        for _ in range(3):
            dist.render_shard(scene, p, r, 8, 0); torch.cuda.synchronize()
            best = min(best, lib.kyhip_kernel_ms(0))
        ts.append(best)
    dist.render_shard(scene, p, 0, 1, 0); torch.cuda.synchronize(); dist.render_shard(scene, p, 0, 1, 0); torch.cuda.synchronize()
    t1 = lib.kyhip_kernel_ms(0)
    print(spp, "N=1 %.2f" % t1, "shards", " ".join("%.2f" % t for t in ts), "sum %.2f" % sum(ts), "max/mean %.4f" % (max(ts) / np.mean(ts)), "eff(max) %.4f eff(mean) %.4f" % (t1 / 8 / max(ts), t1 / 8 / np.mean(ts)))
