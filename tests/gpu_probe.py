"""Ad-hoc GPU/oracle comparison used while bringing the kernels up (not a pytest file)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ky_amd import api, _abi as A
from oracle import kyoracle as O

def rel(a, b):
    return np.abs(a - b) / np.maximum(1e-6, np.maximum(np.abs(a), np.abs(b)))

for name, scene, W, H in (("cornell", api.cornell_box_scene(A.CB_DEFAULT_SCENE, 64, 64), 64, 64), ("veach", api.mis_scene(96, 54), 96, 54)):
    for strat in (A.DIRECT_BOTH_MIS, A.DIRECT_BSDF, A.DIRECT_LIGHT, A.DIRECT_BSDF_MIS, A.DIRECT_LIGHT_MIS, A.DIRECT_IDLE):
        p = api.make_params(W, H, 64, direct_sample=strat)
        bad = 0; tot = 0; maxd = 0
        for (x, y) in ((W//2, H//2), (5, 5), (W//3, 2*H//3), (W-4, H-3)):
            g = api.kat_li(scene, p, x, y, 0, 256)
            c = O.li(scene, p, x, y, 0, 256)
            d = np.abs(g - c).max(axis=1); s = np.maximum(1e-3, np.abs(c).max(axis=1))
            bad += int((d / s > 1e-3).sum()); tot += 256; maxd = max(maxd, float((d/s).max()))
        t = time.time(); g = api.render(scene, p); tg = time.time() - t
        t = time.time(); c = O.render(scene, p); tc = time.time() - t
        print(f"{name} strat {strat:2d}: per-sample mismatches {bad}/{tot} (max rel {maxd:.2e})  film RMSE {np.sqrt(((g-c)**2).mean()):.2e} "
              f"mean g {g.mean():.5f} c {c.mean():.5f}  kernel {api.kernel_ms():.2f} ms  gpu wall {tg:.3f}s  cpu {tc:.3f}s")
# big one
scene = api.cornell_box_scene(A.CB_DEFAULT_SCENE, 1024, 768)
for spp in (16, 64):
    p = api.make_params(1024, 768, spp)
    t = time.time(); g = api.render(scene, p); tg = time.time() - t
    ms = api.kernel_ms()
    print(f"cornell 1024x768x{spp}: kernel {ms:.2f} ms -> {1024*768*spp/ms/1e3:.1f} Msamples/s (wall {tg:.2f}s) mean {g.mean():.5f}")
scene = api.mis_scene(1280, 720)
p = api.make_params(1280, 720, 16)
g = api.render(scene, p); ms = api.kernel_ms()
print(f"veach 1280x720x16: kernel {ms:.2f} ms -> {1280*720*16/ms/1e3:.1f} Msamples/s mean {g.mean():.5f}")
