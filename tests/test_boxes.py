"""Boxes: axis-aligned rectangles that are whole faces of one box are tested by the nearest-hit traversal with one slab test.
Which surfaces qualify is host code (find_boxes, checked here without a GPU); that the slab test finds what scene_t::intersect's scan finds
(ky.cpp:3172-3184) is checked on the GPU: against the per-rectangle traversal of the same library (kyhip_set_boxes(0)) and against the oracle."""
import os

import numpy as np
import pytest

from helpers import CustomScene, make_light, make_material, make_shape, prove_ties, random_rays, rmse, unit


def _rect(A, axis, c, lo, hi, flip=False):
    """an axis-aligned rectangle in the plane x_axis = c spanning [lo, hi] on the other two axes (cyclic order)"""
    u, v = (axis + 1) % 3, (axis + 2) % 3
    pts = []
    for a, b in ((lo[0], lo[1]), (hi[0], lo[1]), (hi[0], hi[1]), (lo[0], hi[1])):
        p = [0.0, 0.0, 0.0]
        p[axis], p[u], p[v] = c, a, b
        pts.append(tuple(p))
    return make_shape(A, A.SHAPE_RECTANGLE, pts, flip=flip)


def _box_faces(A, lo, hi, faces=(0, 1, 2, 3, 4, 5)):
    """the faces (2 axis + side) of the box [lo, hi] as rectangles"""
    out = []
    for f in faces:
        axis, side = f // 2, f % 2
        u, v = (axis + 1) % 3, (axis + 2) % 3
        out.append(_rect(A, axis, hi[axis] if side else lo[axis], (lo[u], lo[v]), (hi[u], hi[v]), flip=bool(side)))
    return out


def _scene(A, api, shapes, lights=(), light_of=None):
    cam = A.Camera.from_buffer_copy(api.cornell_box_scene(A.CB_DEFAULT_SCENE, 8, 8).c.camera)
    mats = [make_material(A, A.MATERIAL_MATTE, (0.5, 0.5, 0.5))]
    surfaces = [A.Surface(i, 0, -1 if light_of is None else light_of.get(i, -1)) for i in range(len(shapes))]
    return CustomScene(A, cam, list(shapes), mats, list(lights), surfaces)


def test_boxes_of_the_shipped_scenes(A, api):
    lib = A.load_kyhip()
    area = api.cornell_box_scene(A.CB_BOTH_SMALL_SPHERES | A.CB_LIGHT_AREA, 64, 64)
    # surfaces (3514-3529 order): left, right, top, bottom, back, the two balls, then the lamp housing left2, right2, front2, back2 and the lamp bottom2
    n, faces = api.scene_boxes(area)
    assert n == 2
    assert faces.tolist() == [0, 1, 5, 4, 2, -1, -1, 8 + 0, 8 + 1, 8 + 3, 8 + 2, 8 + 4]   # the room is open towards the camera (+y), the housing towards the ceiling
    for flag in (A.CB_LIGHT_POINT, A.CB_LIGHT_DIRECTION, A.CB_LIGHT_ENVIRONMENT):
        n, faces = api.scene_boxes(api.cornell_box_scene(A.CB_BOTH_SMALL_SPHERES | flag, 64, 64))
        assert n == 1 and faces.tolist() == [0, 1, 5, 4, 2, -1, -1]
    n, faces = api.scene_boxes(api.mis_scene(96, 54))     # Veach: a floor, a back wall and four tilted planks
    assert n == 0 and (faces == -1).all()
    for switch in (lib.kyhip_set_boxes, lib.kyhip_set_specialisation):
        prev = switch(0)
        try:
            assert api.scene_boxes(area)[0] == 0
        finally:
            switch(prev)
    assert api.scene_boxes(area)[0] == 2


def test_scene_facts(A, api):
    """kyhip_scene_facts: what decides the instantiation a launch takes (ky_scene.hpp's KY_FEAT_*)."""
    lib = A.load_kyhip()
    area = api.cornell_box_scene(A.CB_BOTH_SMALL_SPHERES | A.CB_LIGHT_AREA, 64, 64)
    # one area light (1) that samples a rectangle (2), few carriers (4), small tables (128), the lamp is its own carrier (256), boxes (512), all planar surfaces axis rectangles (1024),
    # the one plastic surface (the floor) a rectangle (2048)
    assert api.scene_facts(area) == 1 + 2 + 4 + 128 + 256 + 512 + 1024 + 2048
    assert api.scene_facts(api.cornell_box_scene(A.CB_BOTH_SMALL_SPHERES | A.CB_LIGHT_POINT, 64, 64)) == 2 + 4 + 8 + 128 + 512 + 1024 + 2048   # (no area light: 2 and 4 hold vacuously)
    assert api.scene_facts(api.mis_scene(96, 54)) == 4 + 32 + 64 + 128 + 2048 + 4096           # sphere lamps, no delta lobes; its planks are tilted (no 1024) about the x axis (4096)
    prev = lib.kyhip_set_boxes(0)
    try:
        assert api.scene_facts(area) == 1 + 2 + 4 + 128 + 256 + 1024 + 2048
    finally:
        lib.kyhip_set_boxes(prev)
    prev = lib.kyhip_set_specialisation(0)
    try:
        assert api.scene_facts(area) == 0
    finally:
        lib.kyhip_set_specialisation(prev)
    # a tilted wall among axis rectangles: no 1024
    c, s = np.cos(0.3), np.sin(0.3)
    tilted = make_shape(A, A.SHAPE_RECTANGLE, [(0, 0, 0), (c, s, 0), (c, s, 1), (0, 0, 1)])
    facts = api.scene_facts(_scene(A, api, _box_faces(A, (-1, -1, 0), (1, 1, 2), (0, 1, 2, 3, 4)) + [tilted]))
    assert facts & 512 and not facts & 1024, facts


def test_box_rules(A, api):
    lo, hi = (-1.0, -0.5, 0.0), (1.0, 0.75, 2.0)
    ball = make_shape(A, A.SHAPE_SPHERE, [(0, 0, 1)], radius=0.3)

    def boxes(shapes):
        n, faces = api.scene_boxes(_scene(A, api, shapes))
        return n, faces.tolist()

    assert boxes(_box_faces(A, lo, hi) + [ball]) == (1, [0, 1, 2, 3, 4, 5, -1])                      # a closed box
    assert boxes(_box_faces(A, lo, hi, (5, 0, 3, 2))) == (1, [5, 0, 3, 2])                           # four faces, any order, any orientation of the stored normals
    assert boxes(_box_faces(A, lo, hi, (0, 2, 4))) == (0, [-1, -1, -1])                              # a corner: three rectangle tests are cheaper than a slab test
    # a face must be the WHOLE face: a wall that stops short of the ceiling is no face (the other four still are)
    short = _rect(A, 0, lo[0], (lo[1], lo[2]), (hi[1], 1.5))
    assert boxes([short] + _box_faces(A, lo, hi, (1, 2, 3, 4))) == (1, [-1, 1, 2, 3, 4])
    # any other axis-aligned rectangle in the plane of a FACE: exact ties are decided by list order (3177-3180), which a box cannot reproduce
    poster = _rect(A, 0, lo[0], (0.0, 0.5), (0.5, 1.0))
    assert boxes(_box_faces(A, lo, hi, (0, 1, 2, 3, 4)) + [poster])[0] == 0
    assert boxes(_box_faces(A, lo, hi, (0, 1, 2, 3, 4)) + _box_faces(A, lo, hi, (0,)))[0] == 0         # the same face twice
    # ... but the plane of an OPEN side is nobody's: a lamp housing may reach the ceiling of the room it hangs in
    room = _box_faces(A, (-2, -2, 0), (2, 2, 2), (0, 1, 2, 3, 4, 5))
    housing = _box_faces(A, (-0.25, -0.25, 1.9), (0.25, 0.25, 2.0), (0, 1, 2, 3, 4))
    n, faces = boxes(room + housing)
    assert n == 2 and faces == [0, 1, 2, 3, 4, 5, 8, 9, 10, 11, 12]
    # rectangles that are not axis-aligned are never faces
    c, s = np.cos(0.3), np.sin(0.3)
    turned = [make_shape(A, A.SHAPE_RECTANGLE, [tuple(np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]]) @ np.array(p)) for p in
                                               [tuple(sh.p[k][j] for j in range(3)) for k in range(4)]]) for sh in _box_faces(A, lo, hi)]
    assert boxes(turned)[0] == 0
    # a face's sorted surface index travels in four bits of the hit distance: a box behind fifteen other rectangles is left to the rectangle scan
    many = [_rect(A, 0, -5.0 - 0.1 * k, (0, 0), (1, 1)) for k in range(15)]
    assert boxes(many + _box_faces(A, lo, hi, (2, 3, 4, 5)))[0] == 0


def _rays_for(scene_c, rng, n):
    """rays from inside and outside the scene's bounding region, a share of them aimed at the edges and corners of the Cornell room"""
    rays = random_rays(rng, n, origin_box=2.0, target=rng.uniform(-1.3, 1.3, (n, 3)), tmax_inf_fraction=0.8)
    X0, X1, Y0, Y1, Z0, Z1 = -1.27029, 1.28975, -1.30455, 1.25549, -1.28002, 1.28002
    k = n // 8
    corners = np.array([[x, y, z] for x in (X0, X1) for y in (Y0, Y1) for z in (Z0, Z1)], np.float32)
    o = rng.uniform(-1.0, 1.0, (k, 3)).astype(np.float32)
    tgt = corners[rng.integers(0, 8, k)].copy()
    free = rng.integers(0, 3, k)                      # one coordinate free: a point on an edge; jittered by a few ulp
    tgt[np.arange(k), free] = rng.uniform(-1.2, 1.2, k)
    tgt += rng.normal(0, 2e-7, (k, 3)).astype(np.float32)
    rays[:k, 0:3] = o
    rays[:k, 3:6] = unit(tgt - o)
    rays[:k, 6] = np.inf
    return rays


@pytest.mark.gpu
def test_slab_test_finds_what_the_rectangle_scan_finds(A, api, O):
    lib = A.load_kyhip()
    rng = np.random.default_rng(77)
    lo, hi = (-1.0, -0.5, 0.0), (1.0, 0.75, 2.0)
    scenes = {
        "cornell_area": api.cornell_box_scene(A.CB_BOTH_SMALL_SPHERES | A.CB_LIGHT_AREA, 64, 64),
        "cornell_point": api.cornell_box_scene(A.CB_BOTH_SMALL_SPHERES | A.CB_LIGHT_POINT, 64, 64),
        "four_faces_and_a_ball": _scene(A, api, _box_faces(A, lo, hi, (5, 0, 3, 2)) + [make_shape(A, A.SHAPE_SPHERE, [(0, 0, 1)], radius=0.3)]),
    }
    for name, scene in scenes.items():
        assert api.scene_boxes(scene)[0] >= 1, name
        rays = _rays_for(scene, rng, 8192)
        with_boxes = api.kat_scene_intersect(scene, rays)
        prev = lib.kyhip_set_boxes(0)
        try:
            assert api.scene_boxes(scene)[0] == 0
            scan = api.kat_scene_intersect(scene, rays)
        finally:
            lib.kyhip_set_boxes(prev)
        same = (with_boxes[:, 0] == scan[:, 0]) & (with_boxes[:, 8] == scan[:, 8])
        hit = same & (scan[:, 0] > 0)
        # the slab test's distance carries the surface in its last four mantissa bits: 15 units in the last place, 1.8e-6 (box_update_nearest)
        assert np.abs(with_boxes[hit, 1] - scan[hit, 1]).max() <= 4e-6 * np.maximum(1.0, scan[hit, 1]).max(), name
        assert np.abs(with_boxes[hit, 2:5] - scan[hit, 2:5]).max() <= 2e-5, name
        assert np.abs(with_boxes[hit, 5:8] - scan[hit, 5:8]).max() <= 1e-6, name
        # where the two disagree about the surface the ray passes an edge of a box: the oracle itself gives the slab test's answer for a ray moved by 3e-5
        differ = np.flatnonzero(~same)
        # (the first eighth of the rays is AIMED at edges: a quarter of those are decided by the last bits either way; of the random rest hardly any)
        assert (differ >= len(rays) // 8).sum() <= 0.002 * len(rays), (name, differ.size, int((differ >= len(rays) // 8).sum()))
        if differ.size:
            def oracle_answer(row):
                h = O.kat_scene_intersect(scene, row[None, :])[0]
                return (float(h[0]), float(h[8]))

            def moved(row, g):
                r = row.copy()
                r[0:3] += g.normal(0, 3e-5, 3).astype(np.float32)
                return r
            prove_ties([(int(i), rays[i]) for i in differ], lambda i: (float(with_boxes[i, 0]), float(with_boxes[i, 8])), oracle_answer, moved, rng, "box edge (%s)" % name)


@pytest.mark.gpu
def test_images_with_and_without_boxes(A, api):
    """The switch moves an image by what 15 units in the last place of a hit distance are worth: measured here, bounded at a few times that."""
    lib = A.load_kyhip()
    worst = 0.0
    for flag, spp in ((A.CB_LIGHT_AREA, 256), (A.CB_LIGHT_POINT, 64), (A.CB_LIGHT_ENVIRONMENT, 64)):
        scene = api.cornell_box_scene(A.CB_BOTH_SMALL_SPHERES | flag, 96, 96)
        p = api.make_params(96, 96, spp)
        on = api.render(scene, p)
        kernel = lib.kyhip_last_kernel(0).decode()
        prev = lib.kyhip_set_boxes(0)
        try:
            off = api.render(scene, p)
            kernel_off = lib.kyhip_last_kernel(0).decode()
        finally:
            lib.kyhip_set_boxes(prev)
        if flag == A.CB_LIGHT_AREA and "render_kernel<strategy 48" in kernel:
            # 1 + 2 + 4 + 128 + 256 (the lamp kernel's facts) + 512 (boxes) + 1024 (every planar surface an axis rectangle) + 2048 (plastic on rectangles only); without boxes the 1415 row
            assert "feat 3975" in kernel and "feat 1415" in kernel_off, (kernel, kernel_off)
        fin = np.isfinite(on) & np.isfinite(off)
        d = np.abs(on - off)[fin]
        # a sample whose path takes another discrete decision (a shadow ray at its threshold) moves its pixel by up to 1 / spp of the clamp range: few of them
        print("boxes on / off", flag, spp, "rmse %.3g" % rmse(on[fin], off[fin]), "max %.3g" % float(d.max()), "values beyond 2e-3:", int((d > 2e-3).sum()), "beyond 2e-5:", int((d > 2e-5).sum()), "of", d.size)
        # measured (256 spp, the lamp kernel): 27 of 27 648 values beyond 2e-5, 6 beyond 2e-3, rmse 3.8e-4 -- the single flipped samples that carry the lamp's radiance
        assert rmse(on[fin], off[fin]) <= 1e-3, (flag, rmse(on[fin], off[fin]))
        assert (d > 2e-3).sum() <= 24 and (d > 2e-5).sum() <= 0.005 * d.size, (flag, int((d > 2e-3).sum()), int((d > 2e-5).sum()), float(d.max()))
        worst = max(worst, float(d.max()))
    assert worst > 0.0   # (the switch did switch something)


def _boxy_room(A, api, seed, W, H):
    """A room nobody tuned for: an axis-aligned room (five or six walls), one or two axis-aligned crates standing in it (four to six faces each), a sphere or two,
    every second seed a tilted panel (a parallelogram that is no axis rectangle: no KY_FEAT_AXIS_ALIGNED), and ONE light -- a point light or a rectangle lamp under
    the ceiling (the kernels with the box traversal are the one-light ones).  The camera looks in through the open side or from inside."""
    g = np.random.default_rng(9000 + seed)
    half = g.uniform(1.0, 1.6, 3)
    lo, hi = tuple(-half), tuple(half)
    shapes = _box_faces(A, lo, hi, (0, 1, 2, 4, 5) if seed % 3 else (0, 1, 2, 3, 4, 5))          # open towards +y for two seeds of three
    for _ in range(int(g.integers(1, 3))):
        c = g.uniform(-0.6, 0.6, 3)
        c[2] = -half[2]                                                                        # standing on the floor
        ext = g.uniform(0.15, 0.4, 3)
        faces = [f for f in range(6) if f != 4]                                                # no bottom face
        if g.random() < 0.5:
            faces.remove(int(g.choice([0, 1, 2, 3])))                                          # a side missing: four faces
        shapes += _box_faces(A, (c[0] - ext[0], c[1] - ext[1], c[2]), (c[0] + ext[0], c[1] + ext[1], c[2] + 2 * ext[2]), tuple(faces))
    for _ in range(int(g.integers(1, 3))):
        r = float(g.uniform(0.15, 0.3))
        p = g.uniform(-0.7, 0.7, 3)
        shapes.append(make_shape(A, A.SHAPE_SPHERE, [(float(p[0]), float(p[1]), float(-half[2] + r + g.uniform(0.0, 0.8)))], radius=r))
    if seed % 2:
        a = float(g.uniform(0.2, 0.6))
        shapes.append(make_shape(A, A.SHAPE_RECTANGLE, [(-0.5, 0.2, 0.0), (0.0, 0.2 + a, 0.0), (0.0, 0.2 + a, 0.5), (-0.5, 0.2, 0.5)]))
    mats = [make_material(A, A.MATERIAL_MATTE, tuple(g.uniform(0.3, 0.8, 3))), make_material(A, A.MATERIAL_MIRROR, (0.9, 0.9, 0.9)),
            make_material(A, A.MATERIAL_PLASTIC, (0.2, 0.2, 0.2), (0.6, 0.6, 0.6), exponent=40.0)]
    lights, light_of = [], {}
    if seed % 4 < 2:
        lights.append(make_light(A, A.LIGHT_POINT, (3, 3, 3), position=(0.1, -0.2, float(half[2] - 0.3))))
    else:
        z = float(half[2] - 0.05)
        shapes.append(make_shape(A, A.SHAPE_RECTANGLE, [(-0.3, -0.3, z), (-0.3, 0.3, z), (0.3, 0.3, z), (0.3, -0.3, z)]))
        lights.append(make_light(A, A.LIGHT_AREA, (12, 12, 12), shape=len(shapes) - 1))
        light_of[len(shapes) - 1] = 0
    cam = A.Camera.from_buffer_copy(api.cornell_box_scene(A.CB_DEFAULT_SCENE, W, H).c.camera)
    if seed % 3 == 0:                                                                          # closed room: the camera inside, at the +y wall looking in
        cam.position[0], cam.position[1], cam.position[2] = 0.0, float(half[1] - 0.05), 0.1
    is_sphere = [sh.kind == A.SHAPE_SPHERE for sh in shapes]
    surfaces = [A.Surface(i, (1 if (is_sphere[i] and i % 2) else (2 if i == 3 else 0)), light_of.get(i, -1)) for i in range(len(shapes))]
    return CustomScene(A, cam, shapes, mats, lights, surfaces)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(int(os.environ.get("KY_BOXY_ROOMS", "8"))))   # KY_BOXY_ROOMS=64: the soak profiles/r05_boxy_room_soak.txt reports
def test_boxy_rooms_nobody_tuned_for(A, api, O, seed):
    """The box traversal on scenes it was not written against: several boxes with four to six faces, rays from inside and outside, a camera inside a closed room,
    with and without a tilted panel (KY_FEAT_AXIS_ALIGNED), a point light or a lamp -- the kernels' rows with KY_FEAT_BOXES against the oracle."""
    lib = A.load_kyhip()
    W, H = 48, 40
    scene = _boxy_room(A, api, seed, W, H)
    n_box, faces = api.scene_boxes(scene)
    assert n_box >= 2, (seed, n_box)
    facts = api.scene_facts(scene)
    assert facts & 512 and bool(facts & 1024) == (seed % 2 == 0), (seed, facts)
    rng = np.random.default_rng(100 + seed)
    rays = random_rays(rng, 4096, origin_box=2.0, target=rng.uniform(-1.2, 1.2, (4096, 3)), tmax_inf_fraction=0.8)
    g, c = api.kat_scene_intersect(scene, rays), O.kat_scene_intersect(scene, rays)
    same = (g[:, 0] == c[:, 0]) & (g[:, 8] == c[:, 8])
    hit = same & (c[:, 0] > 0)
    assert (~same).sum() <= 0.003 * len(rays), (seed, int((~same).sum()))
    assert np.abs(g[hit, 1] - c[hit, 1]).max() <= 2e-5 * max(1.0, float(c[hit, 1].max())) and np.abs(g[hit, 2:5] - c[hit, 2:5]).max() <= 5e-5, seed

    def oracle_answer(row):
        h = O.kat_scene_intersect(scene, row[None, :])[0]
        return (float(h[0]), float(h[8]))

    def moved(row, gen):
        r = row.copy()
        r[0:3] += gen.normal(0, 3e-5, 3).astype(np.float32)
        return r
    prove_ties([(int(i), rays[i]) for i in np.flatnonzero(~same)], lambda i: (float(g[i, 0]), float(g[i, 8])), oracle_answer, moved, rng, "boxy room %d" % seed)
    # the film through the render kernel the facts select
    p = api.make_params(W, H, 128, tile_w=16, tile_h=8)
    film, ref = api.render(scene, p), O.render(scene, p)
    kernel = lib.kyhip_last_kernel(0).decode()
    if "render_kernel<strategy 48" in kernel:
        import re
        row = int(re.search(r"feat (\d+)", kernel).group(1))
        assert row & ~facts == 0, (seed, facts, kernel)                     # a row assumes only what holds
        # the rows with the box traversal: one point / directional light (8 + 512 + 1024 + 2048), or the lamp kernel with small tables (... + 128); a room with a tilted
        # panel or more than 16 surfaces under a lamp runs a row without boxes
        if seed % 2 == 0 and (seed % 4 < 2 or facts & 128):
            assert row & 512, (seed, facts, kernel)
    fin = np.isfinite(ref).all(axis=2)
    assert np.isfinite(film).all() and fin.mean() > 0.995 and ref.mean() > 0.002, (seed, float(ref.mean()))
    e = rmse(film[fin], ref[fin])
    note = ""
    if e >= 1e-3:
        # one sample of 128 that takes another discrete decision than the oracle's and carries the lamp's radiance is 2e-3 of RMSE on this frame (soak room 14): the pixels that
        # are off are set aside only if every differing sample of theirs is explained (helpers.explain_pixel asserts it)
        from helpers import rmse_with_explained_flips
        e_without, e_with, n_exempt = rmse_with_explained_flips(api, O, scene, p, film, ref, max_exempt=8, threshold=5e-3)
        note = " (%.2e with %d explained pixel(s))" % (e_with, n_exempt)
        e = e_without
    print("boxy room %d: %d boxes, facts %d, kernel %s, film RMSE %.2e%s" % (seed, n_box, facts, kernel.split(" =")[0], e, note))
    assert e < 1e-3, (seed, e, kernel)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(int(os.environ.get("KY_BOXY_ROOMS", "8"))))
def test_boxy_rooms_any_pair(A, api, O, seed):
    """The any-hit form of the slab test (box_update_any, round 6: the environment estimate's pair scan) on the rooms nobody tuned for -- two or three boxes with four to
    six faces each, rays from inside, outside and from a hair off the faces themselves, with and without an end: "meets a surface" against the oracle's
    scene_t::intersect, every disagreement a proven tie."""
    W, H = 48, 40
    scene = _boxy_room(A, api, seed, W, H)
    n_box, _ = api.scene_boxes(scene)
    assert n_box >= 2
    rng = np.random.default_rng(700 + seed)
    n, box = 4096, 1.8
    first = O.kat_scene_intersect(scene, random_rays(rng, 3 * n, origin_box=box, target=rng.uniform(-box, box, (3 * n, 3))))
    first = first[first[:, 0] == 1][:n]
    on = first[:, 2:5] + 1e-2 * first[:, 5:8]
    free = rng.uniform(-box, box, (len(on), 3))
    n = len(on)
    oa, ob = np.where(rng.uniform(size=(n, 1)) < 0.6, on, free), np.where(rng.uniform(size=(n, 1)) < 0.6, on, free)
    da, db = unit(rng.normal(size=(n, 3))), unit(rng.normal(size=(n, 3)))
    tb = rng.uniform(0.05, 3.0 * box, n)
    rows = np.concatenate([oa, da, ob, db, tb[:, None]], 1).astype(np.float32)
    g = api.kat_any_pair(scene, rows)
    ra = np.concatenate([rows[:, 0:6], np.full((n, 1), np.inf, np.float32)], 1)
    rb = np.concatenate([rows[:, 6:12], rows[:, 12:13]], 1)
    ca, cb = O.kat_scene_intersect(scene, ra)[:, 0], O.kat_scene_intersect(scene, rb)[:, 0]
    assert (g[:, 0] != ca).mean() < 4e-3 and (g[:, 1] != cb).mean() < 4e-3, (seed, (g[:, 0] != ca).mean(), (g[:, 1] != cb).mean())
    eps = 3e-5
    for col, rays, c in ((0, ra, ca), (1, rb, cb)):
        for i in np.flatnonzero(g[:, col] != c):
            found = False
            for _ in range(64):
                r = rays[i].copy()
                r[0:3] += (eps * box) * rng.uniform(-1, 1, 3).astype(np.float32)
                r[3:6] = unit(r[3:6] + eps * rng.uniform(-1, 1, 3)).astype(np.float32)
                if np.isfinite(r[6]):
                    r[6] *= 1 + eps * rng.uniform(-1, 1)
                if O.kat_scene_intersect(scene, r[None])[0, 0] == g[i, col]:
                    found = True
                    break
            assert found, ("the pair scan differs from the oracle away from any tie", seed, col, rays[i], g[i], c[i])
