"""N=1 and N=8 shard kernel times (Cornell 1024x768x1024spp) of the loaded library: the scaling efficiency at kernel level."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ky_amd import api, dist, _abi as A
lib = A.load_kyhip()
scene = api.cornell_box_scene(A.CB_DEFAULT_SCENE, 1024, 768)
p = api.make_params(1024, 768, 1024)
def t(r, N):
    best = 1e9
    for _ in range(2):
        dist.render_shard(scene, p, r, N, 0); torch.cuda.synchronize(); best = min(best, lib.kyhip_kernel_ms(0))
    return best
full = t(0, 1)
sh = [t(r, 8) for r in (0, 3, 7)]
print("N=1 %.2f ms   N=8 shards %s   efficiency %.3f" % (full, " ".join("%.2f" % x for x in sh), full / (8 * max(sh))))
