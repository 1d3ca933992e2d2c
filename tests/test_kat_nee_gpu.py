"""Function-level known-answer test of the six direct-lighting estimators' halves (SURVEY 8(c) `kat_nee_*`): for vertices found by
tracing random rays into both shipped scenes -- so positions, normals, wo and surfaces are real path vertices -- and random numbers
shared with the oracle, each light's BSDF-sampling half and light-sampling half must agree term by term.  A term is either equal
within tolerance or zero on exactly one side: a discrete outcome that differs (the shadow ray's occlusion at the self-occlusion
threshold, quirk 1; whether the BSDF-sampled ray reaches the light); those are counted and bounded."""
import numpy as np
import pytest

from helpers import random_rays

pytestmark = pytest.mark.gpu


def _vertices(A, api, O, scene, rng, n, box):
    rays = random_rays(rng, 4 * n, origin_box=box, target=rng.uniform(-box, box, (4 * n, 3)))
    hit = O.kat_scene_intersect(scene, rays)
    ok = hit[:, 0] == 1
    rays, hit = rays[ok][:n], hit[ok][:n]
    m = len(hit)
    rows = np.zeros((m, 15), np.float32)
    rows[:, 0:3] = hit[:, 2:5]                    # position
    rows[:, 3:6] = hit[:, 5:8]                    # normal as the shape reports it
    rows[:, 6:9] = -rays[:, 3:6]                  # wo = -ray direction (3125)
    rows[:, 9] = hit[:, 8]                        # surface
    rows[:, 10:15] = rng.uniform(0, 1, (m, 5))    # lobe number, random_bsdf, random_light
    return rows


def _compare(g, c, value_tol):
    """-> (terms compared, flips).  g, c: [n, 3] one half of the estimate."""
    # "zero": below 1e-30 -- a Phong value far from its peak is a denormal on the CPU and flushed to zero by the GPU's fp32 mode
    gz, cz = (np.abs(g).max(axis=1) < 1e-30), (np.abs(c).max(axis=1) < 1e-30)
    flips = int((gz != cz).sum())
    both = ~gz & ~cz
    if both.any():
        scale = np.maximum(np.abs(c[both]).max(axis=1, keepdims=True), 1e-6)
        rel = np.abs(g[both] - c[both]) / scale
        assert np.quantile(rel.max(axis=1), 0.999) < value_tol, float(rel.max())
    return int((~(gz & cz)).sum()), flips


@pytest.mark.parametrize("which", ["cornell_area", "cornell_point", "cornell_direction", "cornell_environment", "veach"])
def test_direct_lighting_estimators_term_by_term(which, A, api, O, rng):
    if which == "veach":
        scene, box, n_lights, tol = api.mis_scene(64, 36), 6.0, 5, 2e-2     # exponent-5000 lobe: pow amplifies its base's rounding
    else:
        flag = {"cornell_area": A.CB_LIGHT_AREA, "cornell_point": A.CB_LIGHT_POINT, "cornell_direction": A.CB_LIGHT_DIRECTION,
                "cornell_environment": A.CB_LIGHT_ENVIRONMENT}[which]
        scene, box, n_lights, tol = api.cornell_box_scene(A.CB_BOTH_SMALL_SPHERES | flag, 64, 64), 1.2, 1, 5e-4
    rows = _vertices(A, api, O, scene, rng, 4096, box)
    assert len(rows) > 2000
    terms = flips = 0
    for light in range(n_lights):
        for strategy in (A.DIRECT_BSDF, A.DIRECT_LIGHT, A.DIRECT_BSDF_MIS, A.DIRECT_LIGHT_MIS, A.DIRECT_BOTH_MIS):
            g, c = api.kat_nee(scene, strategy, light, rows), O.kat_nee(scene, strategy, light, rows)
            assert np.isfinite(g).all() or not np.isfinite(c).all()
            fin = np.isfinite(c).all(axis=1) & np.isfinite(g).all(axis=1)
            for half in (slice(0, 3), slice(3, 6)):
                t, f = _compare(g[fin][:, half], c[fin][:, half], tol)
                terms += t
                flips += f
            if strategy in (A.DIRECT_BSDF, A.DIRECT_BSDF_MIS):
                assert not g[:, 3:6].any() and not c[:, 3:6].any()     # these strategies have no light-sampling half
            if strategy in (A.DIRECT_LIGHT, A.DIRECT_LIGHT_MIS):
                assert not g[:, 0:3].any() and not c[:, 0:3].any()
    print("%s: %d non-zero terms, %d differ in their discrete outcome" % (which, terms, flips))
    assert terms > 500 or which in ("cornell_point", "cornell_direction")
    # measured: Cornell (all four lights) 0 of 1431 ... 7464 terms; Veach 36 of 7668 (the sphere lights' self-occlusion threshold)
    assert flips <= (0.01 if which == "veach" else 0.002) * max(terms, 1), (flips, terms)
