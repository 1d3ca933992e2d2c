"""Where a kyhip_render call's time goes beyond its kernel: wall clock of the call at several spp, the kernel's own duration, and the
pinned-copy rate of the box for scale.  usage: python tools/seam_trace.py"""
import time, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ky_amd import _abi as A, api
W, H = 1024, 768
scene = api.cornell_box_scene(A.CB_DEFAULT_SCENE, W, H)
film = np.zeros((H, W, 3), np.float32)
for spp in (1, 8, 64, 1024):
    p = api.make_params(W, H, spp)
    api.render(scene, p, film=film)
    reps = 20 if spp < 1024 else 3
    t0 = time.perf_counter()
    for _ in range(reps):
        api.render(scene, p, film=film)
    ms = (time.perf_counter() - t0) / reps * 1e3
    print("spp %5d  call %.3f ms  kernel %.3f ms  rest %.3f ms" % (spp, ms, api.kernel_ms(0), ms - api.kernel_ms(0)))
d = torch.zeros(W * H * 3, dtype=torch.float32, device="cuda")
h = torch.zeros(W * H * 3, dtype=torch.float32).pin_memory()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    h.copy_(d, non_blocking=True); torch.cuda.synchronize()
print("pinned D2H of the film: %.3f ms" % ((time.perf_counter() - t0) / 20 * 1e3))
a = np.zeros(W * H * 3, np.float32); b = np.ones(W * H * 3, np.float32)
t0 = time.perf_counter()
for _ in range(20):
    a += b
print("numpy film += film (1 thread): %.3f ms" % ((time.perf_counter() - t0) / 20 * 1e3))
