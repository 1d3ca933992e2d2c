"""The reference's own values for the formulas ky.cpp inherited from smallpt2pbrt/smallpt_rewrite.cpp.

tests/golden/rewrite_kat.npz was produced by the reference's classes themselves (oracle/rewrite_kat.cpp #includes
smallpt_rewrite.cpp unmodified; generator tests/golden/make_rewrite_kat.py, which lists formula by formula what ky.cpp kept and
what it changed).  Those values are fp64; ky.cpp computes the same formulas in fp32, so the CPU restatement (oracle/ky_oracle.cpp)
must agree to fp32 rounding -- 1e-6 relative to the magnitude of the quantity, a few ulp -- and the HIP path (hardware rcp / rsq /
sqrt, FMA contraction) to the tolerance of its own KATs.  This carries the reference pin from SURVEY rows (f)4 to rows a3, a7,
a10, a12, a13: frame_t, sphere_t::intersect, camera_t, the Lambert lobe's value and pdf, the mirror lobe, gamma_encoding -- and, since
round 5, to a21 (areal_radiance: an area light's radiance is one-sided, seen through surface_t::intersect) and a5 (scene_t::intersect:
the whole list is scanned with a shrinking distance and the EARLIER of two surfaces at exactly the same distance stays).
"""
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def G():
    return np.load(os.path.join(HERE, "golden", "rewrite_kat.npz"))


def rel(a, b, floor=1.0):
    """max |a - b| / max(floor, |b|)"""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(floor, np.abs(b)))) if a.size else 0.0


def sphere_shape(A, row):
    s = A.Shape()
    s.kind = A.SHAPE_SPHERE
    for j in range(3):
        s.p[0][j] = float(row[j])
    s.radius = float(row[3])
    return s


def matte(A, R):
    m = A.Material()
    m.kind = A.MATERIAL_MATTE
    for j in range(3):
        m.color0[j] = float(R[j])
    return m


def mirror(A, R):
    m = A.Material()
    m.kind = A.MATERIAL_MIRROR
    for j in range(3):
        m.color0[j] = float(R[j])
    return m


def check_spheres(G, A, kat_intersect, tol):
    """One KAT call per row (the harness built one Sphere per row).  t = neg_b -+ sqrt(discr) with discr = neg_b^2 - oc.oc + r^2: at
    grazing incidence the discriminant cancels and its rounding (in ANY fp32 evaluation, the reference's included) is divided by
    2 sqrt(discr), so the well-conditioned rows (sqrt(discr) >= r / 5: four fifths of the hits) carry the tight bound and the
    rest 10x that."""
    sin, sout = G["sphere_in"], G["sphere_out"].copy()
    out = np.zeros((len(sin), 8), np.float32)
    for i, row in enumerate(sin):
        out[i] = kat_intersect(sphere_shape(A, row), row[None, 4:11].astype(np.float32))[0]
    assert np.array_equal(out[:, 0], sout[:, 0].astype(np.float32)), "hit flags differ from the reference's"
    h = sout[:, 0] == 1
    assert h.sum() > 500
    c, r, o, d = (sin[:, 0:3].astype(np.float64), sin[:, 3].astype(np.float64), sin[:, 4:7].astype(np.float64), sin[:, 7:10].astype(np.float64))
    oc = c - o
    nb = (oc * d).sum(1)
    sq = np.sqrt(np.maximum(nb * nb - (oc * oc).sum(1) + r * r, 0))
    scale = 1.0 + np.linalg.norm(oc, axis=1)
    err_t = np.abs(out[:, 1] - sout[:, 1]) / scale
    err_p = np.abs(out[:, 2:5] - sout[:, 2:5]).max(1) / scale
    err_n = np.abs(out[:, 5:8] - sout[:, 5:8]).max(1) * r / scale     # (p - c) / |p - c| amplifies the position's rounding by 1 / r
    err = np.maximum(np.maximum(err_t, err_p), err_n)
    good = h & (sq >= 0.2 * r)
    assert good.sum() > 0.7 * h.sum()
    assert err[good].max() < tol, err[good].max()
    assert err[h].max() < 10 * tol, err[h].max()


def check_camera(G, api, A, kat_camera, tol):
    scenes = [api.cornell_box_scene(A.CB_DEFAULT_SCENE, 256, 256), api.cornell_box_scene(A.CB_DEFAULT_SCENE, 1024, 768), api.mis_scene(1280, 720)]
    for i, scene in enumerate(scenes):
        assert list(G["cam_res"][i]) == [int(scene.c.camera.resolution[0]), int(scene.c.camera.resolution[1])]
        out = kat_camera(scene.c.camera, G["cam_pfilm"][i])
        assert rel(out[:, 3:6], G["cam_out"][i]) < tol, rel(out[:, 3:6], G["cam_out"][i])


def check_bsdfs(G, A, kat_bsdf, tol):
    bin_, bout = G["bsdf_in"], G["bsdf_out"]
    n = len(bin_)
    cos_o = (bin_[:, 0:3] * bin_[:, 3:6]).sum(1)
    cos_i = (bin_[:, 0:3] * bin_[:, 6:9]).sum(1)
    same = cos_o * cos_i > 1e-4
    other = cos_o * cos_i < -1e-4
    assert same.sum() > 100 and other.sum() > 100
    # the KAT takes one material per call and R is per row in the fixture: call per row group of equal R would be n calls; instead use
    # R = 1 and scale (f is linear in R for both lobes: R / pi and R / |cos|)
    x = np.concatenate([bin_[:, 0:3], bin_[:, 3:6], np.full((n, 2), 0.5, np.float32), bin_[:, 6:9], np.zeros((n, 1), np.float32)], 1).astype(np.float32)
    lam = kat_bsdf(matte(A, (1, 1, 1)), x)
    R = bin_[:, 9:12].astype(np.float64)
    # Lambert value: the reference's R / pi wherever wo and wi share a hemisphere; ky.cpp's 0 across hemispheres (2232) is ky's own
    assert rel(lam[same, 8:11] * R[same], bout[same, 0:3]) < tol
    assert np.all(lam[other, 8:11] == 0)
    # Lambert pdf: |cos| / pi in the same hemisphere, 0 across -- both sources
    decided = same | other
    assert rel(lam[decided, 11], bout[decided, 3]) < tol
    mir = kat_bsdf(mirror(A, (1, 1, 1)), x)
    grazing = np.abs(cos_o) < 1e-2          # f = R / |cos|: the quotient amplifies the rounding of the cosine
    ok = ~grazing
    assert rel(mir[ok, 0:3] * R[ok], bout[ok, 4:7], floor=1e-30) < 20 * tol    # relative: the value is unbounded
    assert rel(mir[:, 3:6], bout[:, 7:10]) < tol
    assert np.all(mir[:, 6] == 1) and np.all(bout[:, 10] == 1)


def check_area_light_le(G, A, li_fn):
    """AreaLight::Le through Primitive::Intersect (smallpt_rewrite.cpp:1114-1117, 1135-1146 == ky.cpp:2957-2960, 3077-3088).  Each row is its own scene -- one
    emitting sphere, a black matte material -- seen along one ray: a 1 x 1 film whose camera sits at the ray's origin and looks along its direction, the debug
    sampler (the camera sample is the pixel centre, i.e. exactly `front`), depth 0: Li of that sample is the emission the hit reports, and nothing else."""
    from helpers import CustomScene, make_light, make_material, make_shape
    lin, lout = G["le_in"], G["le_out"]
    lit = dark = missed = 0
    for row, ref in zip(lin, lout):
        c, r, L, o, d = row[0:3], float(row[3]), row[4:7], row[7:10], row[10:13].astype(np.float64)
        up = np.cross(d, [1.0, 0.0, 0.0] if abs(d[0]) < 0.9 else [0.0, 1.0, 0.0])
        up /= np.linalg.norm(up)
        right = np.cross(up, d)
        cam = A.Camera()
        for j in range(3):
            cam.position[j], cam.front[j], cam.right[j], cam.up[j] = float(o[j]), float(row[10 + j]), float(right[j] * 0.1), float(up[j] * 0.1)
        cam.resolution[0] = cam.resolution[1] = 1.0
        scene = CustomScene(A, cam, [make_shape(A, A.SHAPE_SPHERE, [c], radius=r)], [make_material(A, A.MATERIAL_MATTE, (0, 0, 0))],
                            [make_light(A, A.LIGHT_AREA, tuple(float(x) for x in L), shape=0)], [A.Surface(0, 0, 0)])
        p = A.RenderParams(A.INTEGRATOR_PATH_TRACING_ITERATION, 0, A.DIRECT_BOTH_MIS, 1, A.SAMPLER_DEBUG, 1234, 1, 1, 8, 8, 0, 1)
        got = np.asarray(li_fn(scene, p, 0, 0, 0, 1))[0]
        if ref[0] == 1 and ref[2] > 0:          # hit from outside: the light's own radiance, to the last bit (it is copied, not computed)
            assert np.array_equal(got, L.astype(np.float32)) and np.allclose(ref[2:5], L), (row, got, ref)
            lit += 1
        else:                                    # a miss, or the inside of the sphere (dot(normal, wo) < 0): black
            assert np.all(got == 0) and np.all(ref[2:5] == 0), (row, got, ref)
            dark += ref[0] == 1
            missed += ref[0] == 0
    assert lit > 100 and dark > 100 and missed > 30, (lit, dark, missed)


def check_scene_intersect(G, A, kat_scene_intersect, tol):
    """Scene::Intersect over lists of spheres in which two spheres appear twice (smallpt_rewrite.cpp:1184-1197 == ky.cpp:3172-3184): which list entry the
    scan keeps -- on an exact tie the earlier one, never the duplicate -- and the shrinking distance, for every ray of the fixture."""
    from helpers import CustomScene, make_material, make_shape
    ties = 0
    for k in range(3):
        spheres, rays, ref = G["scene%d_spheres" % k], G["scene%d_rays" % k], G["scene%d_out" % k]
        scene = CustomScene(A, A.Camera(), [make_shape(A, A.SHAPE_SPHERE, [sp[0:3]], radius=float(sp[3])) for sp in spheres],
                            [make_material(A, A.MATERIAL_MATTE, (0.5, 0.5, 0.5))], [], [A.Surface(i, 0, -1) for i in range(len(spheres))])
        scene.scene.camera.resolution[0] = scene.scene.camera.resolution[1] = 1.0
        out = np.asarray(kat_scene_intersect(scene, rays.astype(np.float32)))
        assert np.array_equal(out[:, 0], ref[:, 0].astype(np.float32)), "hit flags differ from the reference's (scene %d)" % k
        h = ref[:, 0] == 1
        assert h.sum() > 100
        assert np.array_equal(out[h, 8], ref[h, 2].astype(np.float32)), ("the scan kept another list entry than the reference's", k, np.flatnonzero(out[h, 8] != ref[h, 2]))
        scale = 1.0 + np.linalg.norm(rays[:, 0:3], axis=1)
        assert (np.abs(out[h, 1] - ref[h, 1]) / scale[h]).max() < 10 * tol and (np.abs(out[h, 2:5] - ref[h, 3:6]).max(1) / scale[h]).max() < 10 * tol
        # the doubled spheres: their later copies are never reported, their first entries are
        dup = [i for i in range(len(spheres)) if any(np.array_equal(spheres[i], spheres[j]) for j in range(i))]
        first = [min(j for j in range(len(spheres)) if np.array_equal(spheres[i], spheres[j])) for i in dup]
        assert len(dup) == 2 and not np.isin(ref[h, 2], dup).any() and not np.isin(out[h, 8], dup).any()
        ties += int(np.isin(ref[h, 2], first).sum())
    assert ties > 60, ties


def test_oracle_area_light_le_matches_the_reference(G, O, A):
    check_area_light_le(G, A, O.li)


def test_oracle_scene_intersect_matches_the_reference(G, O, A):
    check_scene_intersect(G, A, O.kat_scene_intersect, 1e-6)


def test_oracle_frame_matches_the_reference(G, O):
    out = O.kat_frame(G["frame_in"])
    assert rel(out[:, 0:9], G["frame_out"][:, 0:9]) < 1e-6
    assert rel(out[:, 9:15], G["frame_out"][:, 9:15], floor=2.0) < 1e-6
    # the |n.x| > 0.99 branch is in the fixture
    assert (np.abs(G["frame_in"][:, 0]) > 0.99).sum() > 5


def test_oracle_sphere_intersect_matches_the_reference(G, O, A):
    check_spheres(G, A, O.kat_intersect, 1e-6)


def test_oracle_camera_matches_the_reference(G, O, api, A):
    check_camera(G, api, A, O.kat_camera, 1e-6)


def test_oracle_lambert_and_mirror_match_the_reference(G, O, A):
    check_bsdfs(G, A, O.kat_bsdf, 1e-6)


def test_gamma_encoding_matches_the_reference(G):
    from oracle import film_writers as FW
    assert np.array_equal(FW.gamma_encoding(G["gamma_in"]), G["gamma_out"])


def test_what_the_reference_changed_is_visible_in_the_fixture(G, O, A):
    """make_rewrite_kat.py's 'differs' rows, shown on data: the rewrite maps the unit square to the disk by polar coordinates, ky.cpp
    concentrically (710-733) -- so sampled directions are NOT comparable -- while the lift z = sqrt(max(0, 1 - x^2 - y^2)) is shared."""
    u, ref = G["lift_in"].astype(np.float64), G["lift_out"]
    polar = np.stack([np.sqrt(u[:, 0]) * np.cos(2 * np.pi * u[:, 1]), np.sqrt(u[:, 0]) * np.sin(2 * np.pi * u[:, 1])], 1)
    assert rel(ref[:, 0:2], polar) < 1e-12
    assert rel(ref[:, 2], np.sqrt(np.maximum(0, 1 - ref[:, 0] ** 2 - ref[:, 1] ** 2))) < 1e-12
    n = len(u)
    x = np.concatenate([np.tile([0, 0, 1], (n, 1)), np.tile([0, 0, 1], (n, 1)), u, np.tile([0, 0, 1], (n, 1)), np.zeros((n, 1))], 1).astype(np.float32)
    s = O.kat_bsdf(matte(A, (1, 1, 1)), x)[:, 3:6]                      # normal = z: world = local up to the frame's axes
    assert np.abs(s[:, 0:2] - ref[:, 0:2]).max() > 0.1                  # a different mapping
    assert rel(s[:, 2], np.sqrt(np.maximum(0, 1 - s[:, 0].astype(np.float64) ** 2 - s[:, 1].astype(np.float64) ** 2)), floor=0.05) < 1e-5


@pytest.mark.gpu
def test_hip_path_matches_the_reference(G, api, A):
    """The same comparison for the HIP path, at its own KAT tolerances (tests/test_parity_gpu.py: geometry 2e-5, values 2e-4)."""
    check_spheres(G, A, api.kat_intersect, 2e-5)
    check_camera(G, api, A, api.kat_camera, 1e-6)
    check_bsdfs(G, A, api.kat_bsdf, 1e-5)
    check_area_light_le(G, A, api.kat_li)
    check_scene_intersect(G, A, api.kat_scene_intersect, 2e-5)
