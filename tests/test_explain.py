"""The explained-flip machinery under test itself (tests/helpers.py: explain_sample, explain_pixel, rmse_with_explained_flips).

Every film comparison of the suite and bench.py's `parity_gate` set aside pixels whose differing samples are "explained": the two traces of the
sample (kyhip_kat_li_trace / kyo_trace_li, one row per vertex, include/kyhip.h) differ first in a recorded discrete decision, or continuously at or after
a vertex that amplifies rounding (specular, Phong, grazing).  A classifier that explains too much would hide a regression, so here it is shown
what it REFUSES: (1) on hand-made traces, a continuous difference on diffuse surfaces is None whatever its size; (2) on the GPU, a render of a scene that is
slightly wrong -- the matte walls' reflectances off by 2 % -- against the oracle's render of the right one is NOT explained: explain_pixel raises, and
rmse_with_explained_flips with it.
"""
import numpy as np
import pytest

from helpers import explain_sample, explain_pixel, rmse_with_explained_flips


def _row(bounces, surface, lobe, pos, n=(0, 0, 1), wo=(0, 0, 1), beta=(1, 1, 1), lo=(0, 0, 0), flags=0, bsdf_bits=0, light_bits=0):
    r = np.zeros(26, np.float32)
    r[0], r[1], r[2] = bounces, surface, lobe
    r[3:6], r[6:9], r[9:12], r[12:15], r[15:18] = pos, n, wo, beta, lo
    r[18:21], r[21], r[22] = (0.3, 0.3, 0.3), 0.3, 0.5
    r[23], r[24], r[25] = flags, bsdf_bits, light_bits
    return r


def _path(lobes=(0, 0, 0)):
    return [_row(k, 3 + k, lobe, (0.1 * k, 0.2, 0.3), lo=(0.05 * k,) * 3, beta=(0.7 ** k,) * 3, light_bits=1) for k, lobe in enumerate(lobes)]


def test_equal_decisions_and_a_continuous_difference_on_diffuse_surfaces_is_not_explained():
    c = _path()
    for what, idx, delta in (("position", 3, 3e-4), ("normal", 7, 3e-4), ("throughput", 13, 1e-3), ("radiance", 16, 1e-3)):
        g = [r.copy() for r in c]
        g[1][idx] += delta
        assert explain_sample(g, c) is None, what
    # the same differences below the tolerances are not differences: equal vertices -> the last traversal decided ("decision")
    g = [r.copy() for r in c]
    g[1][3] += 2e-5
    g[2][16] += 5e-5
    assert explain_sample(g, c) == "decision"


def test_a_decision_that_differs_first_explains_and_one_that_differs_later_does_not():
    c = _path()
    for j in (1, 2, 23, 24, 25):
        g = [r.copy() for r in c]
        g[1][j] += 1
        g[2][3:6] += 0.5            # after a flipped decision the paths are different paths
        assert explain_sample(g, c) == "decision", j
        # ... but a continuous difference BEFORE the flipped decision is what it is
        g[0][13] += 1e-2
        assert explain_sample(g, c) is None, j
    # one path ends earlier (roulette, depth, the last traversal)
    assert explain_sample(c[:2], c) == "decision" and explain_sample(c, c[:1]) == "decision"


def test_amplifiers_count_only_at_or_before_the_first_difference():
    # mirror at vertex 1: a difference at vertex 1 or 2 is "specular", at vertex 0 it is not
    c = _path((0, 1, 0))
    for k, want in ((0, None), (1, "specular"), (2, "specular")):
        g = [r.copy() for r in c]
        g[k][4] += 1e-3
        assert explain_sample(g, c) == want, k
    # Phong lobe at vertex 1: its own radiance is already amplified (the lobe's value is in Lo), geometry at that vertex is not (it came from the vertex before)
    c = _path((0, 3, 0))
    g = [r.copy() for r in c]; g[1][16] += 1e-2
    assert explain_sample(g, c) == "phong"
    g = [r.copy() for r in c]; g[1][4] += 1e-3
    assert explain_sample(g, c) is None
    g = [r.copy() for r in c]; g[2][4] += 1e-3
    assert explain_sample(g, c) == "phong"
    g = [r.copy() for r in c]; g[0][16] += 1e-2
    assert explain_sample(g, c) is None
    # a grazing hit (|cos(normal, wo)| < 0.05) at vertex 1
    c = _path()
    s = np.float32(np.sqrt(1 - 0.03 ** 2))
    c[1][9:12] = (s, 0, 0.03)
    for k, want in ((0, None), (1, "grazing"), (2, "grazing")):
        g = [r.copy() for r in c]
        g[k][4] += 1e-3
        assert explain_sample(g, c) == want, k
    c[1][9:12] = (np.float32(np.sqrt(1 - 0.06 ** 2)), 0, 0.06)     # not grazing enough
    g = [r.copy() for r in c]; g[2][4] += 1e-3
    assert explain_sample(g, c) is None
    # a caller's small sphere (surface 4 = vertex 1)
    c = _path()
    g = [r.copy() for r in c]; g[2][4] += 1e-3
    assert explain_sample(g, c, amplifying_surfaces=(4,)) == "small sphere" and explain_sample(g, c, amplifying_surfaces=(5,)) == "small sphere"
    assert explain_sample(g, c, amplifying_surfaces=(6,)) is None and explain_sample(g, c) is None


def test_wider_tolerances_are_named_in_the_result():
    c = _path()
    g = [r.copy() for r in c]
    g[1][16] += 5e-4
    assert explain_sample(g, c) is None
    assert explain_sample(g, c, value_tol=1e-3) == "within tolerance"     # the caller asked for more slack and the sample used it: said so, counted apart


class _OtherScene:
    """ky_amd.api with the scene swapped on the way to the GPU: the render under test is of `wrong`, whatever scene the checker passes."""

    def __init__(self, api, wrong):
        self._api, self._wrong = api, wrong

    def kat_li(self, scene, *a, **k):
        return self._api.kat_li(self._wrong, *a, **k)

    def kat_li_trace(self, scene, *a, **k):
        return self._api.kat_li_trace(self._wrong, *a, **k)

    def render(self, scene, *a, **k):
        return self._api.render(self._wrong, *a, **k)


@pytest.mark.gpu
def test_a_render_that_is_merely_wrong_is_not_explained(A, api, O):
    """The Cornell box of configs[1] with its matte walls 2 % darker on the GPU side only.  Decisions are the same as the oracle's in nearly every sample
    (reflectances enter roulette through the throughput, so a few do flip) and the radiance differs continuously from the first matte vertex on: the
    classifier must meet samples it cannot explain in every pixel that sees a matte wall, and say so."""
    W, H, spp = 64, 48, 64
    flags = A.CB_BOTH_SMALL_SPHERES | A.CB_LIGHT_AREA
    right, wrong = api.cornell_box_scene(flags, W, H), api.cornell_box_scene(flags, W, H)
    changed = 0
    for i in range(wrong.c.material_count):
        m = wrong.c.materials[i]
        if m.kind == A.MATERIAL_MATTE and max(m.color0) > 0.5:
            for j in range(3):
                m.color0[j] *= 0.98
            changed += 1
    assert changed == 4       # white, red, green, blue
    params = api.make_params(W, H, spp)
    bad = _OtherScene(api, wrong)
    looked = 0
    for (x, y) in ((12, 24), (52, 24), (32, 3), (32, 16), (20, 44), (45, 44)):     # left and right wall, ceiling, back wall, floor
        first = O.trace_li(right, params, x, y, 0)[0]
        # the machinery on the right scene: nothing to explain in this pixel, or explained
        explain_pixel(api, O, right, params, x, y)
        if int(first[2]) != 0:
            continue              # this pixel's first vertex is not on a matte surface (plastic floor, a ball)
        looked += 1
        with pytest.raises(AssertionError, match="nothing that amplifies rounding"):
            explain_pixel(bad, O, right, params, x, y)
    assert looked >= 3
    # film level: the wrong render is off by a percent of most pixels' values, the right one by rounding
    c = O.render(right, params)
    g_ok, g_bad = api.render(right, params), bad.render(right, params)
    ok, _, _ = rmse_with_explained_flips(api, O, right, params, g_ok, c)      # (a flipped sample is worth 25 / 64 in its pixel at this spp: the plain RMSE is not the measure)
    wrong_by = float(np.sqrt(np.mean((g_bad.astype(np.float64) - c) ** 2)))
    assert ok < 1e-4 and wrong_by > 1e-3, (ok, wrong_by)
    # ... and no pixel of it gets set aside: the first one looked at fails
    with pytest.raises(AssertionError, match="nothing that amplifies rounding"):
        rmse_with_explained_flips(bad, O, right, params, g_bad, c, max_exempt=64, threshold=2e-3)
