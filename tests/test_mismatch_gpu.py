"""Every camera sample whose radiance differs between the HIP path and the oracle is explained, vertex by vertex.

The fp32 results are not bit-exact by construction (FMA contraction, hardware rcp / rsq / sin / cos / exp / log, the dual-basis
parallelogram test): a rounding difference can flip a DISCRETE decision -- which surface a ray hits, whether a shadow ray is
occluded, whether a BSDF-sampled ray reaches the light, which way the glass goes, whether roulette ends the path -- and the
rest of that one path is then a different path.  This test proves that this is the ONLY way samples differ: for each
mismatching sample both sides are traced (kyhip_kat_li_trace / kyo_trace_li) and must agree to 1e-4 on position, normal, wo,
throughput and radiance at every vertex BEFORE the first vertex where a recorded decision differs; a sample whose first
difference is continuous (same decisions, different numbers) fails the test.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GEOM = slice(3, 12)      # position, normal, wo
BETA = slice(12, 15)
LO = slice(15, 18)
BS = slice(18, 23)       # bs.f, pdf, |cos|


def _close(a, b, tol):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return bool(np.all(np.abs(a - b) <= tol * np.maximum(1.0, np.maximum(np.abs(a), np.abs(b)))))


def _explain(g_rows, c_rows, g_li, c_li, value_tol, geom_tol_after_phong):
    """-> (kind of the first difference, vertex index); raises AssertionError if the paths differ without a decision differing."""
    n = min(len(g_rows), len(c_rows))
    geom_tol = 1e-4
    for k in range(n):
        g, c = g_rows[k], c_rows[k]
        # what led INTO this vertex must agree: the same surface was hit from the same direction with the same throughput
        if g[1] != c[1]:
            return "nearest-hit flip", k
        assert _close(g[GEOM], c[GEOM], geom_tol), ("geometry differs at vertex %d with identical decisions" % k, g[GEOM], c[GEOM])
        if c[2] == 3:
            # a direction sampled from a Phong lobe: sin(theta) = sqrt(1 - cos^2) with cos = u^(1/(n+1)) rounded to fp32 -- for
            # the exponent 5000 that cancellation leaves the REFERENCE's own sin(theta) with ~1e-3 relative rounding noise,
            # which no other arithmetic can reproduce bit for bit; what follows such a bounce is compared with that slack
            geom_tol = max(geom_tol, geom_tol_after_phong)
        assert _close(g[BETA], c[BETA], value_tol), ("throughput differs at vertex %d with identical decisions" % k, g[BETA], c[BETA])
        if g[2] != c[2]:
            return "lobe flip", k      # cannot happen with a shared stream (same number, same threshold); reported if it does
        if g[24] != c[24]:
            return "bsdf-sampled ray / carrier flip", k
        if g[25] != c[25]:
            return "shadow ray flip", k
        # same decisions at this vertex: its direct lighting and its continuation sample are continuous functions of equal inputs
        scale = max(1.0, float(np.abs(c[LO]).max()))
        assert np.all(np.abs(g[LO].astype(np.float64) - c[LO]) <= value_tol * scale), ("radiance differs at vertex %d with identical decisions" % k, g[LO], c[LO])
        if g[23] != c[23]:
            return "glass branch flip", k
        assert _close(g[BS], c[BS], value_tol), ("bsdf sample differs at vertex %d with identical decisions" % k, g[BS], c[BS])
    if len(g_rows) != len(c_rows):
        return "termination flip (roulette / last hit)", n
    # all recorded vertices agree: the radiance can only differ through the last traversal (a hit / miss or emission side flip)
    scale = max(1.0, float(np.abs(c_li).max()))
    if np.all(np.abs(g_li.astype(np.float64) - c_li) <= value_tol * scale):
        return "agrees", n
    return "last traversal flip", n


def _check_scene(api, O, scene, params, pixels, n, value_tol, max_bad_fraction, geom_tol_after_phong=1e-4):
    kinds = {}
    bad = tot = 0
    for (x, y) in pixels:
        g, c = api.kat_li(scene, params, x, y, 0, n), O.li(scene, params, x, y, 0, n)
        fin = np.isfinite(c).all(1)
        d = np.abs(g - c).max(axis=1)
        sc = np.maximum(1e-3, np.abs(c).max(axis=1))
        mism = np.flatnonzero(fin & (d / sc > 1e-3))
        tot += int(fin.sum())
        bad += len(mism)
        for s in mism:
            g_rows, g_li = api.kat_li_trace(scene, params, x, y, int(s))
            c_rows = O.trace_li(scene, params, x, y, int(s))
            assert np.allclose(g_li, g[s], rtol=1e-6, atol=1e-7)          # the trace kernel walks the same path as kat_li
            kind, k = _explain(g_rows, c_rows, g_li, c[s], value_tol, geom_tol_after_phong)
            if kind == "agrees":
                # every vertex agrees within value_tol (relative to max(1, value)) and so does the radiance, yet the sample is off by more than 1e-3 OF ITS OWN
                # VALUE.  Legitimate in two cases.  (a) value_tol is wider than 1e-3: a Phong vertex (pow amplifies the rounding of its base by the exponent).
                # (b) A sample of small radiance whose path was reflected or refracted by a CURVED specular surface before its last vertex: the ball amplifies
                # the hit point's rounding 10-50x per bounce (test_differences_on_the_specular_spheres_are_amplified_rounding), the last vertex then sits 1e-5
                # beside the oracle's, and a light sample seen at a grazing angle changes by 1e-3 of its (tiny) value for that -- found once in 25 600
                # samples on round 5's streams: pixel (18, 50) sample 254, two bounces off the mirror ball, 2.8e-5 apart at the wall, radiance 0.0059
                # against 0.0059 + 6e-6.  Such a sample must still agree to value_tol ABSOLUTELY (it did, or the kind would not be "agrees").
                phong = value_tol > 1e-3 and (c_rows[:, 2] == 3).any()
                curved_specular = len(c_rows) > 1 and np.isin(c_rows[:-1, 2], (1, 2)).any() and float(np.abs(c[s]).max()) < 0.05
                assert phong or curved_specular, (x, y, int(s), "mismatching sample without a differing vertex", g[s], c[s])
                kind = "continuous, within the Phong lobe's pow amplification" if phong else "continuous, a dim sample after a specular bounce (amplified rounding, within value_tol absolutely)"
            kinds[kind] = kinds.get(kind, 0) + 1
    assert bad <= max_bad_fraction * tot, (bad, tot)
    #  the "dim sample after a specular bounce" way out is capped: it was measured once in 25 600 samples; four in one scene's 5 120 would be a
    # regression of that size hiding in it
    dim = sum(v for k, v in kinds.items() if k.startswith("continuous, a dim sample"))
    assert dim <= max(2, tot // 2560), (dim, tot, kinds)
    return bad, tot, kinds


@pytest.mark.parametrize("strategy", [48, 16, 32, 8, 4])
def test_cornell_mismatches_are_decision_flips(strategy, A, api, O):
    scene = api.cornell_box_scene(A.CB_DEFAULT_SCENE, 64, 64)
    params = api.make_params(64, 64, 512, direct_sample=strategy)
    pixels = [(32, 32), (5, 5), (21, 42), (44, 45), (60, 61), (32, 4), (18, 50), (46, 52), (30, 58), (12, 30)]
    # value tolerance: the Cornell floor's Phong lobe has exponent 90 -- pow amplifies a 1e-7 difference of its base 90-fold
    bad, tot, kinds = _check_scene(api, O, scene, params, pixels, 512, 2e-4, 0.02)
    print("cornell strategy %d: %d of %d samples differ; first differences: %s" % (strategy, bad, tot, kinds))


@pytest.mark.parametrize("depth", [5, 16])
def test_veach_mismatches_are_decision_flips(depth, A, api, O):
    scene = api.mis_scene(96, 54)
    params = api.make_params(96, 54, 512, max_path_depth=depth)
    pixels = [(48, 27), (5, 5), (30, 40), (70, 30), (48, 50), (20, 20), (80, 45), (60, 8), (40, 33), (25, 36)]
    # the planks' lobe has exponent 5000: pow turns the 1e-7 rounding of its base into 5e-4 of the value (the 2e-2 KAT tolerance
    # of tests/test_parity_gpu.py::test_kat_bsdf covers the worst case); decisions are exact either way
    bad, tot, kinds = _check_scene(api, O, scene, params, pixels, 512, 2e-2, 0.03, geom_tol_after_phong=3e-3)
    print("veach depth %d: %d of %d samples differ; first differences: %s" % (depth, bad, tot, kinds))


def test_trace_rows_match_on_agreeing_samples(A, api, O):
    """The two trace facilities record the same thing: on samples that agree, every row agrees."""
    scene = api.cornell_box_scene(A.CB_DEFAULT_SCENE, 64, 64)
    params = api.make_params(64, 64, 64)
    checked = 0
    for (x, y) in ((32, 32), (21, 42), (44, 45)):
        g, c = api.kat_li(scene, params, x, y, 0, 64), O.li(scene, params, x, y, 0, 64)
        for s in range(64):
            if not np.isfinite(c[s]).all() or np.abs(g[s] - c[s]).max() > 1e-4 * max(1.0, np.abs(c[s]).max()):
                continue
            g_rows, _ = api.kat_li_trace(scene, params, x, y, s)
            c_rows = O.trace_li(scene, params, x, y, s)
            assert len(g_rows) == len(c_rows)
            if len(g_rows):
                assert np.array_equal(g_rows[:, [0, 1, 2, 23, 24, 25]], c_rows[:, [0, 1, 2, 23, 24, 25]])
                assert np.allclose(g_rows[:, 3:23], c_rows[:, 3:23], rtol=3e-4, atol=3e-5)
            checked += 1
    assert checked > 150


def test_differences_on_the_specular_spheres_are_amplified_rounding(A, api, O):
    """The third way a sample can differ, found on configs[1]'s full-size frame: pixels on the silhouette of the glass sphere.  A curved
    specular surface amplifies a perturbation of the incoming ray (a refraction through the r = 0.5 ball by 10-50x), and at grazing
    incidence sphere_t::intersect's discriminant b^2 - oc.oc + r^2 (ky.cpp:1365-1367) cancels to ~1e-3 of its terms, so the hit point
    itself carries ~1e-4 of rounding in ANY fp32 evaluation order -- the reference's included.  Two bounces later the path is a
    different path.  Checked here: on those pixels every differing sample either differs first in a recorded decision, or its first
    continuous difference is at or after a vertex on one of the two specular spheres; nowhere else."""
    scene = api.cornell_box_scene(A.CB_DEFAULT_SCENE, 1024, 768)
    params = api.make_params(1024, 768, 1024)
    sphere_surfaces = {5, 6}          # left (mirror) and right (glass) ball in create_cornell_box_scene's surface order (3414-3417)
    kinds = {"decision": 0, "after a specular sphere": 0, "of which at grazing incidence (|n.wo| < 0.2)": 0}
    total = 0
    for (x, y) in ((735, 611), (726, 612), (734, 612), (734, 613), (722, 618), (731, 619)):
        g, c = api.kat_li(scene, params, x, y, 0, 1024), O.li(scene, params, x, y, 0, 1024)
        fin = np.isfinite(c).all(1)
        d = np.abs(g - c).max(axis=1)
        sc = np.maximum(1e-3, np.abs(c).max(axis=1))
        for s in np.flatnonzero(fin & (d / sc > 1e-3)):
            total += 1
            g_rows, g_li = api.kat_li_trace(scene, params, x, y, int(s))
            c_rows = O.trace_li(scene, params, x, y, int(s))
            sphere_seen = grazing = False
            explained = None
            for k in range(min(len(g_rows), len(c_rows))):
                gr, cr = g_rows[k], c_rows[k]
                if gr[1] != cr[1] or gr[2] != cr[2] or gr[23] != cr[23] or gr[24] != cr[24] or gr[25] != cr[25]:
                    explained = "decision"
                    break
                if int(cr[1]) in sphere_surfaces:
                    sphere_seen = True
                    grazing = grazing or abs(float(np.dot(cr[6:9], cr[9:12]))) < 0.2
                if not _close(gr[GEOM], cr[GEOM], 1e-4) or not _close(gr[BETA], cr[BETA], 2e-4):
                    assert sphere_seen, ("continuous difference on a path that never touched a specular sphere", x, y, int(s), k, gr[GEOM], cr[GEOM])
                    explained = "after a specular sphere"
                    break
            if explained is None:
                explained = "after a specular sphere" if sphere_seen and len(g_rows) == len(c_rows) else "decision"
            kinds[explained] += 1
            if explained == "after a specular sphere" and grazing:
                kinds["of which at grazing incidence (|n.wo| < 0.2)"] += 1
    assert total > 20 and kinds["after a specular sphere"] > 0
    print("glass silhouette pixels: %d differing samples: %s" % (total, kinds))
