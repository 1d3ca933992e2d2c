"""Parity on scenes nobody tuned for.  The two shipped scenes exercise a narrow slice of the kernel-selection logic (one area light or
five sphere lights, axis-aligned boxes); here seeded random rooms combine tilted parallelograms, axis-aligned rectangles, spheres,
triangles and disks, the four materials, and one to four lights of all four kinds, so that every render-kernel instantiation is met with
geometry it was not written against: the strategy-specialised kernel, the one with deferred shadow rays (two or more lights, with point /
directional / environment lights among them), the run-time-dispatched one, and the GENERAL variants.  Checked against the oracle per
camera sample and as films."""
import os

import numpy as np
import pytest

from helpers import CustomScene, make_light, make_material, make_shape, rmse, rmse_with_explained_flips, unit

pytestmark = pytest.mark.gpu


def _rot(rng, max_angle):
    """A small random rotation (so that walls become general parallelograms instead of axis-aligned rectangles)."""
    axis = unit(rng.normal(size=3))
    a = rng.uniform(-max_angle, max_angle)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(a) * K + (1 - np.cos(a)) * (K @ K)


def random_room(A, api, O, seed, general, W, H):
    rng = np.random.default_rng(seed)
    R = _rot(rng, 0.0 if seed % 3 == 0 else 0.35)          # every third room stays axis-aligned
    X = lambda p: tuple((R @ np.asarray(p, np.float64)).astype(np.float32))
    cam = api.cornell_box_scene(A.CB_DEFAULT_SCENE, W, H)
    camera = A.Camera.from_buffer_copy(cam.c.camera)        # looks along -y into the room from y = 4.1
    a, b, h = 1.3, 1.3, 1.28
    shapes = [
        make_shape(A, A.SHAPE_RECTANGLE, [X((-a, -b, -h)), X((a, -b, -h)), X((a, b, -h)), X((-a, b, -h))]),      # 0 floor
        make_shape(A, A.SHAPE_RECTANGLE, [X((-a, -b, -h)), X((-a, -b, h)), X((a, -b, h)), X((a, -b, -h))]),      # 1 back wall
        make_shape(A, A.SHAPE_RECTANGLE, [X((-a, -b, h)), X((-a, -b, -h)), X((-a, b, -h)), X((-a, b, h))]),      # 2 left
        make_shape(A, A.SHAPE_RECTANGLE, [X((a, -b, -h)), X((a, -b, h)), X((a, b, h)), X((a, b, -h))]),          # 3 right
        make_shape(A, A.SHAPE_RECTANGLE, [X((a, -b, h)), X((-a, -b, h)), X((-a, b, h)), X((a, b, h))]),          # 4 ceiling
    ]
    materials = [make_material(A, A.MATERIAL_MATTE, tuple(rng.uniform(0.2, 0.8, 3))) for _ in range(3)]
    materials.append(make_material(A, A.MATERIAL_PLASTIC, tuple(rng.uniform(0.05, 0.3, 3)), tuple(rng.uniform(0.4, 0.8, 3)), exponent=float(rng.choice([8, 33, 90, 400]))))
    materials.append(make_material(A, A.MATERIAL_MIRROR, (0.95, 0.95, 0.95)))
    materials.append(make_material(A, A.MATERIAL_GLASS, (1, 1, 1), (1, 1, 1), eta=float(rng.uniform(1.3, 1.7))))
    materials.append(make_material(A, A.MATERIAL_MATTE, (0, 0, 0)))                                                   # 6 emitters' own surface
    surfaces = [A.Surface(0, 3, -1), A.Surface(1, 0, -1), A.Surface(2, 1, -1), A.Surface(3, 2, -1), A.Surface(4, 0, -1)]
    for _ in range(int(rng.integers(1, 4))):                                                                          # spheres
        shapes.append(make_shape(A, A.SHAPE_SPHERE, [X((rng.uniform(-0.8, 0.8), rng.uniform(-0.8, 0.6), rng.uniform(-1.0, 0.2)))], radius=float(rng.uniform(0.15, 0.45))))
        surfaces.append(A.Surface(len(shapes) - 1, int(rng.choice([0, 3, 4, 5])), -1))
    if general:                                                                                                       # a free triangle and a disk
        c = np.array([rng.uniform(-0.7, 0.7), rng.uniform(-0.7, 0.3), rng.uniform(-0.9, 0.0)])
        shapes.append(make_shape(A, A.SHAPE_TRIANGLE, [X(c + (-0.4, 0, -0.2)), X(c + (0.4, 0.1, -0.2)), X(c + (0, 0.2, 0.5))]))
        surfaces.append(A.Surface(len(shapes) - 1, int(rng.choice([1, 3, 4])), -1))
        shapes.append(make_shape(A, A.SHAPE_DISK, [X((rng.uniform(-0.8, 0.8), -1.0, rng.uniform(-0.5, 0.8)))], normal=X(unit(np.array([rng.uniform(-0.3, 0.3), 1.0, rng.uniform(-0.3, 0.3)]))),
                                 radius=float(rng.uniform(0.2, 0.4))))
        surfaces.append(A.Surface(len(shapes) - 1, int(rng.choice([2, 4])), -1))
    lights = []
    kinds = list(rng.permutation(["rect", "sphere", "point", "direction", "environment", "disk" if general else "rect2"]))[: int(rng.integers(1, 5))]
    env = -1
    for k in kinds:
        li = len(lights)
        col = tuple(rng.uniform(0.3, 1.0, 3))
        if k in ("rect", "rect2"):
            cx, cy, s = rng.uniform(-0.6, 0.6), rng.uniform(-0.6, 0.6), rng.uniform(0.15, 0.35)
            z = h - 0.02 - 0.01 * li
            shapes.append(make_shape(A, A.SHAPE_RECTANGLE, [X((cx - s, cy - s, z)), X((cx - s, cy + s, z)), X((cx + s, cy + s, z)), X((cx + s, cy - s, z))]))   # faces down
            lights.append(make_light(A, A.LIGHT_AREA, tuple(np.array(col) * 20), shape=len(shapes) - 1))
            surfaces.append(A.Surface(len(shapes) - 1, 6, li))
        elif k == "sphere":
            shapes.append(make_shape(A, A.SHAPE_SPHERE, [X((rng.uniform(-0.8, 0.8), rng.uniform(-0.8, 0.8), rng.uniform(0.3, 0.9)))], radius=float(rng.uniform(0.05, 0.2))))
            lights.append(make_light(A, A.LIGHT_AREA, tuple(np.array(col) * 30), shape=len(shapes) - 1))
            surfaces.append(A.Surface(len(shapes) - 1, 6, li))
        elif k == "disk":
            # (h - 0.025 - 0.01 li: never the plane of a rectangle lamp.  With h - 0.03 the soak's room 21 had its disk and its rectangle lamp overlapping in ONE
            # plane: over the overlap the two hits tie to the last bit, which lamp a ray meets is decided by rounding, and 1.5 % of the samples of a pixel flip.)
            shapes.append(make_shape(A, A.SHAPE_DISK, [X((rng.uniform(-0.6, 0.6), rng.uniform(-0.6, 0.6), h - 0.025 - 0.01 * li))], normal=X((0, 0, -1)), radius=float(rng.uniform(0.15, 0.3))))
            lights.append(make_light(A, A.LIGHT_AREA, tuple(np.array(col) * 20), shape=len(shapes) - 1))
            surfaces.append(A.Surface(len(shapes) - 1, 6, li))
        elif k == "point":
            lights.append(make_light(A, A.LIGHT_POINT, tuple(np.array(col) * 2), position=X((rng.uniform(-0.8, 0.8), rng.uniform(-0.8, 0.8), rng.uniform(0.2, 1.0)))))
        elif k == "direction":
            lights.append(make_light(A, A.LIGHT_DIRECTION, tuple(np.array(col) * 3), direction=tuple(unit(np.array([rng.uniform(-0.5, 0.5), -1.0, rng.uniform(-1.0, -0.2)])))))
        else:
            lights.append(make_light(A, A.LIGHT_ENVIRONMENT, col))
            env = li
    scene = CustomScene(A, camera, shapes, materials, lights, surfaces, environment_light=env)
    radius = float(O.world_bounding_sphere(scene)[3])      # direction / environment lights: preprocess() (3555-3574)
    for l in scene.lights[: len(lights)]:
        if l.kind in (A.LIGHT_DIRECTION, A.LIGHT_ENVIRONMENT):
            l.world_radius = radius
    return scene, kinds


# the suite runs twelve rooms; KY_RANDOM_ROOMS=72 (with or without KYHIP_JIT=1: every room then runs on kernels compiled for it) is the soak DESIGN section 6 reports
@pytest.mark.parametrize("seed", range(int(os.environ.get("KY_RANDOM_ROOMS", "12"))))
def test_random_room(seed, A, api, O):
    W, H = 48, 40
    general = seed % 2 == 1
    scene, kinds = random_room(A, api, O, 4242 + seed, general, W, H)
    pixels = [(24, 20), (6, 30), (40, 30), (24, 6), (10, 12), (36, 14), (30, 34), (16, 26)]
    rates, films = {}, {}
    for strategy in (A.DIRECT_BOTH_MIS, A.DIRECT_LIGHT_MIS, A.DIRECT_BSDF):
        params = api.make_params(W, H, 128, direct_sample=strategy)
        bad = tot = 0
        for (x, y) in pixels:
            g, c = api.kat_li(scene, params, x, y, 0, 128), O.li(scene, params, x, y, 0, 128)
            fin = np.isfinite(c).all(1)
            d = np.abs(g[fin] - c[fin]).max(axis=1)
            s = np.maximum(1e-3, np.abs(c[fin]).max(axis=1))
            bad += int((d / s > 2e-3).sum())
            tot += int(fin.sum())
        rates[strategy] = bad / tot
        assert bad <= 0.01 * tot, (seed, kinds, strategy, bad, tot)                # measured: at most 0.1 %
    # films through the render kernels themselves (both_mis: the specialised or the deferred-shadow-ray instantiation; light_mis:
    # the run-time-dispatched one; GENERAL variants when the room holds a triangle / disk)
    for strategy in (A.DIRECT_BOTH_MIS, A.DIRECT_LIGHT_MIS):
        p = api.make_params(W, H, 256, direct_sample=strategy, tile_w=16, tile_h=8)
        g, c = api.render(scene, p), O.render(scene, p)
        fin = np.isfinite(c).all(axis=2)
        assert np.isfinite(g).all() and g.min() >= 0 and g.max() <= 1 and fin.mean() > 0.995
        e = rmse(g[fin], c[fin])
        if e >= 1.5e-3:
            # (soak room 21: pixels under the ceiling its lamps hang from, one sample of 256 each taking another discrete decision than the oracle's
            # and carrying several units of radiance into a mean that is clamped per pixel.)  The pixels that are off are set aside only if every
            # differing sample of theirs is explained (helpers.explain_pixel asserts it), and the film without them must meet the bound.
            e_without, e_with, n_exempt = rmse_with_explained_flips(api, O, scene, p, g, c, max_exempt=12, threshold=5e-3)
            print("room %d strategy %d: film RMSE %.2e with, %.2e without %d explained pixel(s)" % (seed, strategy, e_with, e_without, n_exempt))
            e = e_without
        films[strategy] = e
        assert c.mean() > 0.002 and e < 1.5e-3, (seed, kinds, strategy, e)    # measured: 1e-7 .. 7.3e-4 (a flipped bright sample in 1920 pixels at 256 spp)
    # the multi-device entry and the kernel choice do not show in the image
    p = api.make_params(W, H, 16)
    assert np.array_equal(api.render(scene, p), api.render_multi(scene, p, [0, 0, 0]))
    # deferred shadow rays (which the library keeps for scenes with many sphere lamps: kyhip_set_shadow_queue) forced on: the same picture up to the
    # order in which a pixel's contributions are summed
    lib = A.load_kyhip()
    for strategy in (A.DIRECT_BOTH_MIS, A.DIRECT_LIGHT_MIS):
        p = api.make_params(W, H, 64, direct_sample=strategy, tile_w=16, tile_h=8)
        inline = api.render(scene, p)
        inline_kernel = lib.kyhip_last_kernel(0)
        prev = lib.kyhip_set_shadow_queue(1)
        try:
            deferred = api.render(scene, p)
            deferred_kernel = lib.kyhip_last_kernel(0)
        finally:
            lib.kyhip_set_shadow_queue(prev)
        assert b"deferred" not in inline_kernel and (general or len(kinds) < 2 or b"deferred shadow rays" in deferred_kernel), (inline_kernel, deferred_kernel)
        # measured: <= 2.4e-7 on rounds 3-5's builds.  The two kernels inline the same expressions into different surroundings, and which multiply-adds the compiler fuses
        # there moves with every edit of the device headers (round 6: code that is dead in both kernels -- the planks' branch of par_coords -- moved one of them): an ulp
        # in a hit distance is a sample that takes another decision once in a few 1e5, i.e. a pixel or two of these 1920 off by (a light's radiance) / spp.  So: every
        # pixel within 2e-6 but at most ten (half a percent), and those within a single sample's weight -- a defect of the deferral itself moves whole regions, not four pixels.
        d = np.abs(inline - deferred).max(axis=2)
        # (one sample's weight: these rooms' lamps carry up to 16 units of radiance, 1 / 64 of which is 0.25 of a clamped pixel; measured 4.4e-4 on room 8, 0.071 on room 10)
        # (the 72-room soak: 0-6 such pixels per room, i.e. a sample in 2e4 takes another decision between the two kernels -- a tenth of what separates either from the oracle)
        assert int((d > 2e-6).sum()) <= 10 and d.max() <= 0.25, (seed, strategy, float(d.max()), int((d > 2e-6).sum()))
    print("room %d (%s%s): mismatching samples both_mis %.2f%% light_mis %.2f%% bsdf %.2f%%; film RMSE both_mis %.2e light_mis %.2e" % (
        seed, "+".join(kinds), ", general shapes" if general else "", 100 * rates[A.DIRECT_BOTH_MIS], 100 * rates[A.DIRECT_LIGHT_MIS], 100 * rates[A.DIRECT_BSDF],
        films[A.DIRECT_BOTH_MIS], films[A.DIRECT_LIGHT_MIS]))
