"""Kernel rate of the Cornell both_mis frame by frame size and depth cap (why the 4096^2 stress frame runs below configs[1]'s rate)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ky_amd import api, _abi as A
for (w, h, depth, spp) in ((1024, 768, 5, 1024), (1024, 768, 16, 1024), (2048, 2048, 5, 192), (4096, 4096, 5, 64), (4096, 4096, 16, 64), (4096, 4096, 16, 256), (512, 512, 5, 4096)):
    scene = api.cornell_box_scene(A.CB_DEFAULT_SCENE, w, h)
    p = api.make_params(w, h, spp, max_path_depth=depth)
    import ctypes as C, torch
    from ky_amd import dist
    lib = A.load_kyhip()
    best = 1e9
    for _ in range(2):
        dist.render_shard(scene, p, 0, 1, 0); torch.cuda.synchronize(); best = min(best, lib.kyhip_kernel_ms(0))
    print("%4dx%4d depth %2d spp %4d: kernel %8.2f ms  %7.1f Msamples/s" % (w, h, depth, spp, best, w * h * spp / best / 1e3), flush=True)
