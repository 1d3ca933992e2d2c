"""The occluder tables: which surfaces a shadow ray never has to test.  The classification is host code (checked here
without a GPU); that leaving those surfaces out never changes scene_t::occluded's answer is checked on the GPU against the full scan."""
import numpy as np
import pytest

from helpers import random_rays, unit


def _scenes(A, api):
    return {
        "cornell_area": api.cornell_box_scene(A.CB_BOTH_SMALL_SPHERES | A.CB_LIGHT_AREA, 64, 64),
        "cornell_point": api.cornell_box_scene(A.CB_BOTH_SMALL_SPHERES | A.CB_LIGHT_POINT, 64, 64),
        "cornell_direction": api.cornell_box_scene(A.CB_BOTH_SMALL_SPHERES | A.CB_LIGHT_DIRECTION, 64, 64),
        "cornell_environment": api.cornell_box_scene(A.CB_BOTH_SMALL_SPHERES | A.CB_LIGHT_ENVIRONMENT, 64, 64),
        "veach": api.mis_scene(96, 54),
    }


def test_classification_of_the_shipped_scenes(A, api):
    S = _scenes(A, api)
    # the Cornell box: left, right, ceiling, floor, back wall have the whole scene on one side ...
    walls = [1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0]
    assert api.scene_non_occluders(S["cornell_area"]).astype(int).tolist() == walls
    # ... for shadow rays towards the ceiling lamp as well (it hangs 2e-2 under the ceiling, more than the ray origin's offset); the
    # lamp's four side panels lie behind its plane and are scanned only for rays that end there (2)
    assert api.scene_non_occluders(S["cornell_area"], 0).astype(int).tolist() == [1, 1, 1, 1, 1, 0, 0, 2, 2, 2, 2, 0]
    for name in ("cornell_point", "cornell_direction", "cornell_environment"):
        assert api.scene_non_occluders(S[name]).astype(int).tolist() == [1, 1, 1, 1, 1, 0, 0]
    assert api.scene_non_occluders(S["cornell_point"], 0).astype(int).tolist() == [1, 1, 1, 1, 1, 0, 0]
    assert not api.scene_non_occluders(S["cornell_direction"], 0).any()      # these rays leave the scene: every surface is tested
    assert not api.scene_non_occluders(S["cornell_environment"], 0).any()
    # Veach: the floor reaches under the back wall and the wall below the floor, the plates are free-standing, the lights are spheres
    assert not api.scene_non_occluders(S["veach"]).any()
    for l in range(5):
        assert not api.scene_non_occluders(S["veach"], l).any()


def test_classification_rules(A, api):
    from helpers import CustomScene, make_light, make_material, make_shape
    cam = A.Camera.from_buffer_copy(api.cornell_box_scene(A.CB_DEFAULT_SCENE, 8, 8).c.camera)
    mats = [make_material(A, A.MATERIAL_MATTE, (0.5, 0.5, 0.5))]
    floor = make_shape(A, A.SHAPE_RECTANGLE, [(-1, -1, 0), (1, -1, 0), (1, 1, 0), (-1, 1, 0)])
    ball_on = make_shape(A, A.SHAPE_SPHERE, [(0, 0, 0.5)], radius=0.5)         # touches the floor's plane
    ball_through = make_shape(A, A.SHAPE_SPHERE, [(0, 0, 0.4)], radius=0.5)    # reaches below it
    tri_up = make_shape(A, A.SHAPE_TRIANGLE, [(0, 0, 1.5), (0.5, 0, 1.5), (0, 0.5, 1.5)], flip=True)   # a triangle light facing down
    panel_above = make_shape(A, A.SHAPE_RECTANGLE, [(0.6, -0.2, 1.5), (0.6, 0.2, 1.5), (0.6, 0.2, 1.8), (0.6, -0.2, 1.8)])
    panel_across = make_shape(A, A.SHAPE_RECTANGLE, [(0.6, -0.2, 1.4), (0.6, 0.2, 1.4), (0.6, 0.2, 1.8), (0.6, -0.2, 1.8)])

    ball_above = make_shape(A, A.SHAPE_SPHERE, [(0, 0, 0.6)], radius=0.5)
    wall = make_shape(A, A.SHAPE_RECTANGLE, [(-1, -1, 0), (-1, 1, 0), (-1, 1, 2), (-1, -1, 2)])       # perpendicular to the floor, touching it
    ramp = make_shape(A, A.SHAPE_RECTANGLE, [(-1, -1, 0), (-1, 1, 0), (0, 1, 1), (0, -1, 1)])         # rises from the floor at 45 degrees

    def left_out(shapes, lights, light=-1, mats_of=None):
        surfaces = [A.Surface(i, 0 if mats_of is None else mats_of[i], -1) for i in range(len(shapes))]
        return api.scene_non_occluders(CustomScene(A, cam, shapes, mats, lights, surfaces), light).astype(int).tolist()

    mats.append(make_material(A, A.MATERIAL_MIRROR, (0.9, 0.9, 0.9)))
    assert left_out([floor, ball_above], []) == [1, 0]
    assert left_out([floor, ball_on], []) == [0, 0]          # shadow rays start 1e-2 off the ball's surface: under the floor's plane, near the contact point
    assert left_out([floor, ball_on], [], mats_of=[0, 1]) == [1, 0]      # ... unless the ball is a mirror: no shadow ray starts there (4571)
    assert left_out([floor, ball_through], []) == [0, 0]
    assert left_out([floor, wall], []) == [1, 1]            # the offset moves along the other plane
    assert left_out([floor, ramp], []) == [0, 0]            # the ramp's offset crosses the floor's plane; the floor's end pokes through the ramp's
    assert left_out([floor, ball_above], [make_light(A, A.LIGHT_POINT, (1, 1, 1), position=(0, 0, 2))]) == [1, 0]
    assert left_out([floor, ball_above], [make_light(A, A.LIGHT_POINT, (1, 1, 1), position=(0, 0, 2))], 0) == [1, 0]
    assert left_out([floor, ball_above], [make_light(A, A.LIGHT_POINT, (1, 1, 1), position=(0, 0, -2))]) == [0, 0]      # a point light under the floor
    # a shadow ray is the segment to the light SHIFTED by the origin's offset (1e-2 along the start surface's normal): a light closer
    # than that to the floor's plane keeps the floor in ITS table
    assert left_out([floor, ball_above], [make_light(A, A.LIGHT_POINT, (1, 1, 1), position=(0, 0, 0.009))], 0) == [0, 0]
    assert left_out([floor, ball_above], [make_light(A, A.LIGHT_POINT, (1, 1, 1), position=(0, 0, 0.009))]) == [1, 0]
    assert left_out([floor, ball_above], [make_light(A, A.LIGHT_POINT, (1, 1, 1), position=(0, 0, 0.011))], 0) == [1, 0]
    tri_light = [make_light(A, A.LIGHT_AREA, (1, 1, 1), shape=2)]
    assert left_out([floor, ball_above, tri_up, panel_above], tri_light) == [1, 0, 0, 0]
    assert left_out([floor, ball_above, tri_up, panel_above], tri_light, 0) == [1, 0, 0, 2]      # mounted behind the triangle light's plane
    assert left_out([floor, ball_above, tri_up, panel_across], tri_light, 0) == [1, 0, 0, 0]     # reaches in front of it
    lamp_on_floor = make_shape(A, A.SHAPE_RECTANGLE, [(-0.2, -0.2, 0), (0.2, -0.2, 0), (0.2, 0.2, 0), (-0.2, 0.2, 0)])
    on_floor = [make_light(A, A.LIGHT_AREA, (1, 1, 1), shape=2)]
    assert left_out([floor, ball_above, lamp_on_floor], on_floor) == [1, 0, 1]
    assert left_out([floor, ball_above, lamp_on_floor], on_floor, 0) == [0, 0, 0]


def _surface_points(A, api, scene, n, seed):
    """points on the scene's non-delta surfaces (where shadow rays start, 4571), their ray-facing normals"""
    rays = random_rays(np.random.default_rng(seed), n, origin_box=1.2, target=np.random.default_rng(seed + 1).uniform(-1.3, 1.3, (n, 3)), tmax_inf_fraction=1.0)
    h = api.kat_scene_intersect(scene, rays)
    c = scene.c if hasattr(scene, "c") else scene.scene
    delta = np.array([c.materials[c.surfaces[i].material].kind in (A.MATERIAL_MIRROR, A.MATERIAL_GLASS) for i in range(c.surface_count)])
    ok = (h[:, 0] > 0) & ~delta[np.maximum(h[:, 8].astype(int), 0)]
    return h[ok, 2:5], h[ok, 5:8]


@pytest.mark.gpu
def test_rays_that_end_at_scene_points(A, api):
    """The carrier query of by_bsdf: a ray from a surface point (offset origin) that ends exactly where it meets the scene again.  The
    table without the walls gives the full scan's answer.  (kyhip_kat_occluded builds scene_t::occluded's ray -- direction towards the
    target, length |target - p| - 2e-3 from the offset origin -- so the target is placed to make that ray end at the chosen point.)"""
    for name, scene in _scenes(A, api).items():
        p, n = _surface_points(A, api, scene, 60000, 11)
        q, nq = _surface_points(A, api, scene, 60000, 23)
        m = min(len(p), len(q))
        p, n, q, nq = p[:m].astype(np.float64), n[:m].astype(np.float64), q[:m].astype(np.float64), nq[:m].astype(np.float64)
        d = unit(q - p)
        for _ in range(3):   # the origin's side depends on the direction, the direction on the origin
            o = p + 1e-2 * n * np.sign(np.sum(n * d, 1, keepdims=True))
            d = unit(q - o)
        length = np.linalg.norm(q - o, axis=1, keepdims=True)
        frac = np.random.default_rng(3).uniform(0.3, 1.0, (m, 1))      # ends ON a scene point, or earlier
        frac[: m // 2] = 1.0 - 1e-4
        target = p + d * (length * frac + 2e-3)
        # (a grazing start: the side of the offset is a coin toss; a grazing arrival: fp32 cannot tell 1 - 1e-4 of the way from all of it)
        keep = (length[:, 0] > 0.05) & (np.abs(np.sum(n * d, 1)) > 0.05) & (np.abs(np.sum(nq * d, 1)) > 0.05)
        seg = np.concatenate([p, n, target], 1).astype(np.float32)[keep]
        full, culled = api.kat_occluded(scene, seg), api.kat_occluded(scene, seg, table=-1)
        assert len(seg) > 20000 and 0.05 < full.mean() < 0.95, (name, full.mean())
        assert np.array_equal(full, culled), (name, int((full != culled).sum()))


@pytest.mark.gpu
def test_shadow_rays_towards_light_samples(A, api, O):
    """Shadow rays as by_emitter traces them -- from a surface point in front of the lamp towards a sample of the lamp, shift and
    overshoot included: the lamp's table gives the full scan's answer.  Most samples taken from the floor ARE blocked: by the lamp
    itself, which the ray runs into (origin 1e-2 up the floor's normal, length measured from the floor)."""
    scene = _scenes(A, api)["cornell_area"]
    sh = scene.c.shapes[scene.c.lights[0].shape]
    p0, p1, p2 = (np.array([sh.p[k][j] for j in range(3)], np.float64) for k in range(3))
    nl = np.array([sh.normal[j] for j in range(3)], np.float64)
    p, n = _surface_points(A, api, scene, 120000, 5)
    u = np.random.default_rng(9).uniform(size=(len(p), 2))
    u[::4] = np.round(u[::4])      # every fourth sample on the lamp's border: where the shifted ray slips past the lamp
    u[::8, 0] = np.random.default_rng(10).uniform(size=len(u[::8]))
    q = p1 + (p0 - p1) * u[:, :1] + (p2 - p1) * u[:, 1:]
    wi = unit(q - p)
    front = (wi @ nl) < 0            # light_sample_Li: Li is black otherwise and no ray is traced
    seg = np.concatenate([p, n, q], 1).astype(np.float32)[front]
    full, culled = api.kat_occluded(scene, seg), api.kat_occluded(scene, seg, table=0)
    assert front.sum() > 40000 and 0.02 < full.mean() < 0.9
    assert np.array_equal(full, culled), int((full != culled).sum())
    floor = np.abs(seg[:, 2] + 1.28) < 1e-3
    assert floor.sum() > 5000 and full[floor].mean() > 0.8
    # the point light of the same box
    scene = _scenes(A, api)["cornell_point"]
    p, n = _surface_points(A, api, scene, 60000, 15)
    lp = np.array([scene.c.lights[0].position[j] for j in range(3)], np.float64)
    seg = np.concatenate([p, n, np.broadcast_to(lp, p.shape)], 1).astype(np.float32)
    full, culled = api.kat_occluded(scene, seg), api.kat_occluded(scene, seg, table=0)
    assert 0.02 < full.mean() < 0.9 and np.array_equal(full, culled), int((full != culled).sum())
    # the random rooms of test_random_scenes_gpu.py: tilted walls, several lights of every kind
    from test_random_scenes_gpu import random_room
    for seed in range(6):
        room, kinds = random_room(A, api, O, 4242 + seed, seed % 2 == 1, 48, 40)
        p, n = _surface_points(A, api, room, 30000, 40 + seed)
        q, _ = _surface_points(A, api, room, 30000, 60 + seed)
        m = min(len(p), len(q))
        for l in range(room.scene.light_count):
            L = room.lights[l]
            if L.kind == A.LIGHT_POINT:
                tq = np.broadcast_to(np.array([L.position[j] for j in range(3)], np.float32), p[:m].shape)
            elif L.kind == A.LIGHT_AREA:
                lsurf = [i for i in range(room.scene.surface_count) if room.surfaces[i].area_light == l]
                h = api.kat_scene_intersect(room, random_rays(np.random.default_rng(70 + seed), 40000, origin_box=1.0, target=np.random.default_rng(71 + seed).uniform(-1.3, 1.3, (40000, 3)), tmax_inf_fraction=1.0))
                on = h[np.isin(h[:, 8].astype(int), lsurf) & (h[:, 0] > 0), 2:5]
                if len(on) < 50:
                    continue
                tq = on[np.random.default_rng(72).integers(0, len(on), m)]
            else:
                continue
            seg = np.concatenate([p[:m], n[:m], tq], 1).astype(np.float32)
            full, culled = api.kat_occluded(room, seg), api.kat_occluded(room, seg, table=l)
            assert np.array_equal(full, culled), (seed, kinds, l, int((full != culled).sum()))


def test_classification_entry_rejects_bad_arguments(A, api):
    import ctypes as C
    lib = A.load_kyhip()
    scene = api.cornell_box_scene(A.CB_BOTH_SMALL_SPHERES | A.CB_LIGHT_AREA, 64, 64)
    with pytest.raises(api.KyError, match="out of range"):
        api.scene_non_occluders(scene, light=1)                       # the scene has one light
    with pytest.raises(api.KyError, match="out of range"):
        api.scene_non_occluders(scene, light=-2)
    out = (C.c_int32 * 4)()
    assert lib.kyhip_scene_non_occluders(scene.flat, -1, out, 4) == A.KY_ERR_INVALID_VALUE        # room for 4 of 12 surfaces
    assert b"12 surfaces" in lib.kyhip_last_error()
    assert lib.kyhip_scene_non_occluders(None, -1, out, 4) == A.KY_ERR_INVALID_VALUE
    assert lib.kyhip_scene_non_occluders(scene.flat, -1, None, 12) == A.KY_ERR_INVALID_VALUE
    # a scene that does not validate is reported as such, not classified
    bad = api.cornell_box_scene(A.CB_BOTH_SMALL_SPHERES | A.CB_LIGHT_AREA, 64, 64)
    bad.c.surfaces[3].material = 99
    big = (C.c_int32 * 64)()
    assert lib.kyhip_scene_non_occluders(bad.flat, -1, big, 64) == A.KY_ERR_INVALID_VALUE
