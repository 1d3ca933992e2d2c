"""kernel time vs samples per pixel (full Cornell frame): the intercept is the per-launch fixed cost (start-up + end-of-queue tail)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ky_amd import api, _abi as A
scene = api.cornell_box_scene(A.CB_DEFAULT_SCENE, 1024, 768)
xs, ys = [], []
for spp in (1, 2, 4, 8, 16, 32, 64, 128, 136, 160, 192, 256, 512):
    p = api.make_params(1024, 768, spp)
    api.render(scene, p)
    ms = []
    for _ in range(3):
        api.render(scene, p); ms.append(api.kernel_ms())
    xs.append(spp); ys.append(min(ms))
    print("spp %4d  kernel %.3f ms  per spp %.4f" % (spp, min(ms), min(ms) / spp), flush=True)
b, a = np.polyfit(xs[-5:], ys[-5:], 1)
print("fit over the last five: %.4f ms/spp, intercept %.3f ms" % (b, a))
