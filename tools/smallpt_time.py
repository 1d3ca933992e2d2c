"""smallpt fp64 path: GPU kernel time vs the CPU oracle on the same box, BASELINE configs[0] and a larger frame."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ky_amd import api
from oracle import kyoracle as O
sp, spo = api.smallpt_scene(), O.smallpt_scene()
for (w, h, samps, cpu) in ((256, 256, 16, True), (1024, 768, 64, False), (1024, 768, 1250, False)):
    p = api.smallpt_params(w, h, samps)
    api.smallpt_render(sp, p)
    t = time.time(); g = api.smallpt_render(sp, p); wall = time.time() - t
    ms = api.kernel_ms()
    n = w * h * 4 * samps
    line = "%dx%d %d spp: kernel %.2f ms (%.1f Msamples/s), call %.1f ms, mean %.5f" % (w, h, 4 * samps, ms, n / ms / 1e3, wall * 1e3, g.mean())
    if cpu:
        t = time.time(); c = O.smallpt_render(spo, p, 0); ct = time.time() - t
        line += " | CPU oracle %.2f s (%.2f Msamples/s, %d threads), rmse %.2e" % (ct, n / ct / 1e6, os.cpu_count(), np.sqrt(((g - c) ** 2).mean()))
    print(line, flush=True)
