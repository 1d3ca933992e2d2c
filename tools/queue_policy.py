"""Deferred shadow rays or not, by number and kind of lights: kernel ms of both_mis on the library's table kernels for the shipped two scenes, the Cornell box with lamp
and point light, and a room with N lights of one kind (N = 1 .. 8; sphere lamps / rectangle lamps / point lights).  KYHIP_SHADOW_QUEUE=0 / 1 forces the choice
(read once per process: run the tool once per setting).   usage: KYHIP_SHADOW_QUEUE=0|1 tools/queue_policy.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from ky_amd import api, _abi as A
from helpers import CustomScene, make_light, make_material, make_shape

lib = A.load_kyhip()


def room(n, kind, W, H):
    return mixed_room({kind: n}, W, H)


def mixed_room(counts, W, H):
    rng = np.random.default_rng(77)
    cam = api.cornell_box_scene(A.CB_DEFAULT_SCENE, W, H)
    camera = A.Camera.from_buffer_copy(cam.c.camera)
    a, b, h = 1.3, 1.3, 1.28
    shapes = [
        make_shape(A, A.SHAPE_RECTANGLE, [(-a, -b, -h), (a, -b, -h), (a, b, -h), (-a, b, -h)]),
        make_shape(A, A.SHAPE_RECTANGLE, [(-a, -b, -h), (-a, -b, h), (a, -b, h), (a, -b, -h)]),
        make_shape(A, A.SHAPE_RECTANGLE, [(-a, -b, h), (-a, -b, -h), (-a, b, -h), (-a, b, h)]),
        make_shape(A, A.SHAPE_RECTANGLE, [(a, -b, -h), (a, -b, h), (a, b, h), (a, b, -h)]),
        make_shape(A, A.SHAPE_RECTANGLE, [(a, -b, h), (-a, -b, h), (-a, b, h), (a, b, h)]),
        make_shape(A, A.SHAPE_SPHERE, [(-0.5, 0.0, -0.8)], radius=0.45),
        make_shape(A, A.SHAPE_SPHERE, [(0.55, 0.1, -0.85)], radius=0.4),
    ]
    materials = [make_material(A, A.MATERIAL_MATTE, (0.7, 0.7, 0.7)), make_material(A, A.MATERIAL_MATTE, (0.7, 0.2, 0.2)), make_material(A, A.MATERIAL_MATTE, (0.2, 0.7, 0.2)),
                 make_material(A, A.MATERIAL_PLASTIC, (0.1, 0.1, 0.1), (0.7, 0.7, 0.7), exponent=90.0), make_material(A, A.MATERIAL_MIRROR, (0.95, 0.95, 0.95)),
                 make_material(A, A.MATERIAL_GLASS, (1, 1, 1), (1, 1, 1), eta=1.5), make_material(A, A.MATERIAL_MATTE, (0, 0, 0))]
    surfaces = [A.Surface(0, 3, -1), A.Surface(1, 0, -1), A.Surface(2, 1, -1), A.Surface(3, 2, -1), A.Surface(4, 0, -1), A.Surface(5, 4, -1), A.Surface(6, 5, -1)]
    lights = []
    order = [k for k in ("sphere", "rect", "point") for _ in range(counts.get(k, 0))]
    for li, kind in enumerate(order):
        x, y = -0.9 + 1.8 * (li % 4) / 3.0, -0.5 + 0.45 * (li // 4)
        if kind == "sphere":
            shapes.append(make_shape(A, A.SHAPE_SPHERE, [(x, y, 0.9)], radius=0.08))
            lights.append(make_light(A, A.LIGHT_AREA, (30, 30, 30), shape=len(shapes) - 1)); surfaces.append(A.Surface(len(shapes) - 1, 6, li))
        elif kind == "rect":
            z, s = h - 0.02 - 0.001 * li, 0.12
            shapes.append(make_shape(A, A.SHAPE_RECTANGLE, [(x - s, y - s, z), (x - s, y + s, z), (x + s, y + s, z), (x + s, y - s, z)]))
            lights.append(make_light(A, A.LIGHT_AREA, (20, 20, 20), shape=len(shapes) - 1)); surfaces.append(A.Surface(len(shapes) - 1, 6, li))
        else:
            lights.append(make_light(A, A.LIGHT_POINT, (2, 2, 2), position=(x, y, 0.9)))
    return CustomScene(A, camera, shapes, materials, lights, surfaces, environment_light=-1)


def ms(scene, p):
    api.render(scene, p)
    best = 1e9
    for _ in range(3):
        api.render(scene, p); best = min(best, api.kernel_ms())
    return best, lib.kyhip_last_kernel(0).decode().replace("render_kernel", "").replace(", integrator 11", "")


print("KYHIP_SHADOW_QUEUE =", os.environ.get("KYHIP_SHADOW_QUEUE"))
print("veach 1280x720x256          %8.3f ms  %s" % ms(api.mis_scene(1280, 720), api.make_params(1280, 720, 256)))
two = api.cornell_box_scene(A.CB_BOTH_SMALL_SPHERES | A.CB_LIGHT_AREA | A.CB_LIGHT_POINT, 1024, 768)
print("cornell lamp+point x256     %8.3f ms  %s" % ms(two, api.make_params(1024, 768, 256)))
if "mixed" in sys.argv:
    for c in ({"sphere": 5, "point": 1}, {"sphere": 6, "point": 1}, {"sphere": 7, "point": 2}, {"sphere": 5, "rect": 1}, {"sphere": 5, "rect": 3}, {"sphere": 4, "rect": 4},
              {"sphere": 6, "rect": 2, "point": 2}, {"sphere": 8, "rect": 2, "point": 2}, {"sphere": 3, "rect": 3, "point": 2}):
        print("room, %-36s x128   %8.3f ms  %s" % ((str(c),) + ms(mixed_room(c, 640, 480), api.make_params(640, 480, 128))), flush=True)
else:
    for kind in ("sphere", "rect", "point"):
        for n in (1, 2, 3, 4, 5, 6, 8):
            print("room, %d %-6s lights x128   %8.3f ms  %s" % ((n, kind) + ms(room(n, kind, 640, 480), api.make_params(640, 480, 128))), flush=True)
