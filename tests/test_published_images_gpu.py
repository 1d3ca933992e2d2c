"""The reference's published renders (docs/images, README.md:15-29) as a parity gate, where they can be one.

tests/golden/reference_images.npz holds 16 x 16 block means of the 8-bit, gamma-encoded mosaics the reference's drivers wrote
(generator: tests/golden/make_image_fixtures.py).  Each cell is ONE low-spp render; its expectation is estimated here by
averaging K independent GPU renders of the same spp after the same clamp + gamma.  The JPEG mosaics come from an OLDER revision
of the source than /root/reference/ky.cpp (e.g. delta lights are not halved under both_mis there, cf. ky.cpp:3977 / 4083, and the
fixture itself shows it: see the last assertion), so only the cells whose code path did not change since are gated: the `idle` and
`light`-strategy cells, the black cells, and the environment-light cells."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_images.npz")
K = 16


def _expected_blocks(api, scene, W, H, spp, strat, block=16):
    acc = np.zeros((H, W, 3), np.float64)
    for k in range(K):
        f = api.render(scene, api.make_params(W, H, spp, direct_sample=strat, seed=1000 + k))
        acc += np.clip(f, 0, 1) ** (1 / 2.2)
    acc /= K
    h, w = (H // block) * block, (W // block) * block
    return acc[:h, :w].reshape(h // block, block, w // block, block, 3).mean(axis=(1, 3))


def test_veach_mosaic_cells(A, api):
    """render_mis_scene (ky.cpp:4878-4905): cells bsdf, light, idle of the first mosaic row (the second row is not aligned to the
    16-pixel blocks of the fixture)."""
    ref = np.load(GOLD)["veach_mis"]
    scene = api.mis_scene(512, 308)
    idle = np.abs(_expected_blocks(api, scene, 512, 308, 10, A.DIRECT_IDLE) - ref[0:19, 64:96])
    assert idle.mean() < 1e-3 and idle.max() < 0.03, (idle.mean(), idle.max())          # measured 2e-4 / 0.017: emitters seen directly, JPEG ringing
    light = np.abs(_expected_blocks(api, scene, 512, 308, 10, A.DIRECT_LIGHT) - ref[0:19, 32:64])
    assert light.mean() < 0.03, light.mean()                                               # measured 0.018 (one 10-spp sample of a noisy estimator + JPEG)


def test_cornell_mosaic_cells(A, api):
    """render_multiple_scene (ky.cpp:4819-4876): rows bsdf / light / both_mis, columns point / direction / area / environment."""
    ref = np.load(GOLD)["multi_scene_mis"]
    cols = [(A.CB_LIGHT_POINT, 10), (A.CB_LIGHT_DIRECTION, 40), (A.CB_LIGHT_AREA, 40), (A.CB_LIGHT_ENVIRONMENT, 10)]
    gated = {(0, 0): 1e-6, (0, 1): 1e-6,      # bsdf strategy under a delta light: black (3894)
             (0, 3): 0.02,                    # bsdf, environment       measured 0.012
             (1, 0): 0.03, (1, 1): 0.02,      # light, point / direction 0.017 / 0.009
             (2, 3): 0.015}                   # both_mis, environment    0.0065
    for (r, c), bound in gated.items():
        strat = (A.DIRECT_BSDF, A.DIRECT_LIGHT, A.DIRECT_BOTH_MIS)[r]
        flag, spp = cols[c]
        scene = api.cornell_box_scene(A.CB_BOTH_SMALL_SPHERES | flag, 256, 256)
        e = _expected_blocks(api, scene, 256, 256, spp, strat)
        d = np.abs(e - ref[r * 16:(r + 1) * 16, c * 16:(c + 1) * 16])
        assert d.mean() < bound, ((r, c), d.mean())
    # The area-light column, the estimator every BASELINE config runs (both_mis under an area light, 4076-4088).  The published bsdf cell (no
    # shadow rays) matches a CONVERGED render everywhere -- the mosaic was rendered with more samples than today's driver takes (4821-4827) --
    # and so do the WALL blocks (block rows 4-8 of a cell: between the lamp's glow and the spheres) of the light and both_mis cells.  Their
    # ceiling row and floor rows do not: those are the regions quirk 1 darkens (shadow rays from the floor hit the lamp's own rectangle,
    # 3187-3201; SURVEY 8(a) quirk 1 measured it on the reference: 100 of 100 light samples from a floor point occluded), i.e. the published
    # build predates the self-occluding shadow ray.  The walls are therefore a (weak) reference-produced gate on the headline estimator; the
    # ceiling and floor rows are asserted to DIFFER, so that this reading of the fixture stays checked.
    scene = api.cornell_box_scene(A.CB_BOTH_SMALL_SPHERES | A.CB_LIGHT_AREA, 256, 256)

    def converged_rows(strat):
        f = api.render(scene, api.make_params(256, 256, 1024, direct_sample=strat))
        g = (np.clip(f, 0, 1) ** (1 / 2.2)).reshape(16, 16, 16, 16, 3).mean(axis=(1, 3))
        return g

    d = np.abs(converged_rows(A.DIRECT_BSDF) - ref[0:16, 32:48])
    assert d.mean() < 0.02, d.mean()                                                            # CPU oracle at 640 spp: 0.007 ... 0.018 per block row
    for r, strat in ((1, A.DIRECT_LIGHT), (2, A.DIRECT_BOTH_MIS)):
        d = np.abs(converged_rows(strat) - ref[r * 16:(r + 1) * 16, 32:48]).mean(axis=(1, 2))   # per block row
        assert d[4:9].mean() < 0.02, (r, d[4:9])                                                # CPU oracle at 160 spp: 0.011 (light), 0.009 (both_mis)
        assert d[0] > 0.03 and d[12:16].mean() > 0.03, (r, d[0], d[12:16])                      # 0.059 / 0.048 (light), 0.049 / 0.045 (both_mis)
    # why the other cells are not gated -- a fact of the fixture alone: the published both_mis cells under the point and the
    # directional light equal the published light cells (0.526 / 0.195 both), i.e. that build did not yet halve delta lights
    # under both_mis as ky.cpp:4083 does
    cell = lambda r, c: ref[r * 16:(r + 1) * 16, c * 16:(c + 1) * 16].mean()
    assert abs(cell(2, 0) - cell(1, 0)) < 0.005 and abs(cell(2, 1) - cell(1, 1)) < 0.005
