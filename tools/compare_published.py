#!/usr/bin/env python3
"""Compare the GPU path with the images the reference published (block means, tests/golden/reference_images.npz).

Each published cell is ONE low-spp render, clamped per pixel and gamma encoded; its expectation is estimated here by
averaging K independent renders (different seeds) of the same spp after the same clamp + gamma."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ky_amd import api, _abi as A

g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "reference_images.npz"))
K = int(sys.argv[1]) if len(sys.argv) > 1 else 32

def expected_blocks(scene, W, H, spp, strat, block=16):
    acc = np.zeros((H, W, 3), np.float64)
    for k in range(K):
        f = api.render(scene, api.make_params(W, H, spp, direct_sample=strat, seed=1000 + k))
        acc += np.clip(f, 0, 1) ** (1 / 2.2)
    acc /= K
    h, w = (H // block) * block, (W // block) * block
    return acc[:h, :w].reshape(h // block, block, w // block, block, 3).mean(axis=(1, 3))

print("== veach_mis.jpg (render_mis_scene: 2x3 cells of 512x308, 10 spp)")
scene = api.mis_scene(512, 308)
ref = g["veach_mis"]  # (38, 96, 3): rows 0-18 cell row 0 (308 = 19.25 blocks -> row alignment differs for the 2nd row)
strats = [4, 8, 0, 16, 32, 48]
for cell, s in enumerate(strats):
    r, c = divmod(cell, 3)
    e = expected_blocks(scene, 512, 308, 10, s)
    if r == 0:
        rb = ref[0:19, c * 32:(c + 1) * 32]
        d = np.abs(e - rb)
        print("cell", cell, "strategy", s, "mean abs", d.mean().round(4), "p95", np.quantile(d, 0.95).round(4), "max", d.max().round(3), "mean ref", rb.mean().round(4), "mean gpu", e.mean().round(4))
    else:
        print("cell", cell, "strategy", s, "(second mosaic row is not block aligned: compared by mean only) mean gpu", e.mean().round(4))

print("== multi_scene_mis.jpg (render_multiple_scene: 3x4 cells of 256x256; spp 10/40/40/10)")
ref = g["multi_scene_mis"]  # (48, 64, 3)
flags = [(A.CB_LIGHT_POINT, 10), (A.CB_LIGHT_DIRECTION, 40), (A.CB_LIGHT_AREA, 40), (A.CB_LIGHT_ENVIRONMENT, 10)]
for r, s in enumerate([4, 8, 48]):
    for c, (flag, spp) in enumerate(flags):
        scene = api.cornell_box_scene(A.CB_BOTH_SMALL_SPHERES | flag, 256, 256)
        e = expected_blocks(scene, 256, 256, spp, s)
        rb = ref[r * 16:(r + 1) * 16, c * 16:(c + 1) * 16]
        d = np.abs(e - rb)
        print("row", r, "strategy", s, "col", c, "mean abs", d.mean().round(4), "p95", np.quantile(d, 0.95).round(4), "max", d.max().round(3), "mean ref", rb.mean().round(4), "mean gpu", e.mean().round(4))
