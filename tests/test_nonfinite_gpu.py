"""Non-finite radiance through the film path.  The reference adds clamp01(L) per pixel (3726): +inf clamps to 1, and std::clamp
leaves NaN as NaN, which the 8-bit writers store as 0 (DESIGN.md "Non-finite samples").  The GPU keeps its sums in fixed point, so a
chunk sum that is NaN / +inf / -inf (or beyond the accumulator's range) raises a flag bit instead and the pixel resolves to 0 / 1 / 0.
An emitter with such a colour drives every branch of that classification -- in the chunk flush of the render kernels and, with a
second light in the scene, in the deferred shadow rays' resolve."""
import numpy as np
import pytest

from helpers import CustomScene, make_light, make_material, make_shape

pytestmark = pytest.mark.gpu


def _room(A, api, colour, second_light):
    W, H = 48, 40
    camera = A.Camera.from_buffer_copy(api.cornell_box_scene(A.CB_DEFAULT_SCENE, W, H).c.camera)
    shapes = [
        make_shape(A, A.SHAPE_RECTANGLE, [(-1.3, -1.3, -1.28), (1.3, -1.3, -1.28), (1.3, 1.3, -1.28), (-1.3, 1.3, -1.28)]),      # floor
        make_shape(A, A.SHAPE_RECTANGLE, [(-1.3, -1.3, -1.28), (-1.3, -1.3, 1.28), (1.3, -1.3, 1.28), (1.3, -1.3, -1.28)]),      # back wall
        make_shape(A, A.SHAPE_RECTANGLE, [(-0.4, -1.0, 0.2), (-0.4, -1.0, 0.9), (0.4, -1.0, 0.9), (0.4, -1.0, 0.2)]),            # a panel facing the camera
        make_shape(A, A.SHAPE_SPHERE, [(0.6, 0.0, -0.9)], radius=0.35),
    ]
    materials = [make_material(A, A.MATERIAL_MATTE, (0.7, 0.7, 0.7)), make_material(A, A.MATERIAL_MATTE, (0, 0, 0)),
                 make_material(A, A.MATERIAL_PLASTIC, (0.2, 0.2, 0.2), (0.5, 0.5, 0.5), exponent=30.0)]
    lights = [make_light(A, A.LIGHT_AREA, colour, shape=2)]
    if second_light:
        lights.append(make_light(A, A.LIGHT_POINT, (1.0, 1.0, 1.0), position=(0.0, 0.5, 1.0)))
    surfaces = [A.Surface(0, 0, -1), A.Surface(1, 0, -1), A.Surface(2, 1, 0), A.Surface(3, 2, -1)]
    return CustomScene(A, camera, shapes, materials, lights, surfaces), W, H


@pytest.mark.parametrize("second_light", [False, True])
@pytest.mark.parametrize("colour", [(np.nan, np.inf, 1e30), (-np.inf, 4.0, np.nan), (np.inf, -1e30, 0.5)])
def test_nonfinite_emitter(colour, second_light, A, api, O):
    lib = A.load_kyhip()
    prev = lib.kyhip_set_shadow_queue(1 if second_light else -1)   # two lights: through the deferred rays' resolve (the library would trace this room's rays inline)
    try:
        _nonfinite_emitter(colour, second_light, A, api, O, lib)
    finally:
        lib.kyhip_set_shadow_queue(prev)


def _nonfinite_emitter(colour, second_light, A, api, O, lib):
    scene, W, H = _room(A, api, colour, second_light)
    for strategy in (A.DIRECT_BOTH_MIS, A.DIRECT_LIGHT_MIS):
        p = api.make_params(W, H, 64, direct_sample=strategy, tile_w=16, tile_h=8)
        with np.errstate(all="ignore"):
            g, c = api.render(scene, p), O.render(scene, p)
        assert np.isfinite(g).all() and g.min() >= 0 and g.max() <= 1
        assert (b"deferred shadow rays" in lib.kyhip_last_kernel(0)) == second_light, lib.kyhip_last_kernel(0)
        nan = np.isnan(c)
        assert (nan.any() or not np.isnan(colour).any()) and (g[nan] == 0).all()   # NaN stays NaN in the reference; the writers (and the GPU film) make it 0
        fin = ~nan
        assert np.abs(g[fin] - c[fin]).max() < 2e-2 and np.sqrt(np.mean((g[fin] - c[fin]) ** 2)) < 2e-3
        assert (c[fin] == 1).any() and (c[fin] == 0).any()             # +inf / 1e30 clamp to 1, -inf / -1e30 to 0: both occur
    # the same frame in shards: flags are OR-ed and sums added per shard, the picture does not change
    p1 = api.make_params(W, H, 32, tile_w=16, tile_h=8)
    whole = api.render(scene, p1)
    parts = np.zeros_like(whole)
    for k in range(3):
        api.render(scene, api.make_params(W, H, 32, tile_w=16, tile_h=8, tile_first=k, tile_step=3), film=parts)
    assert np.array_equal(whole, parts)
