"""Shared input generators for the parity tests (seeded; the same arrays go to the oracle and to the GPU)."""
import numpy as np


def unit(v):
    return v / np.linalg.norm(v, axis=-1, keepdims=True)


def random_rays(rng, n, origin_box=3.0, tmax_inf_fraction=0.7, target=None):
    """n x {o[3], d[3], tmax}: origins in a box, directions towards `target` points (or random) so many rays hit."""
    o = rng.uniform(-origin_box, origin_box, (n, 3))
    if target is None:
        d = unit(rng.normal(size=(n, 3)))
    else:
        d = unit(target - o)
    tmax = np.where(rng.uniform(size=n) < tmax_inf_fraction, np.inf, rng.uniform(0.0, 6.0, n))
    return np.concatenate([o, d, tmax[:, None]], 1).astype(np.float32)


def hemisphere_dirs(rng, n):
    return unit(rng.normal(size=(n, 3)))


def rmse(a, b):
    return float(np.sqrt(np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2)))


class CustomScene:
    """A ky_scene assembled directly from ctypes structs (the C-ABI view of scene_t's public constructor, ky.cpp:3151)."""

    def __init__(self, A, camera, shapes, materials, lights, surfaces, environment_light=-1):
        import ctypes as C
        self._keep = (shapes, materials, lights, surfaces)
        self.shapes = (A.Shape * len(shapes))(*shapes)
        self.materials = (A.Material * len(materials))(*materials)
        self.lights = (A.Light * max(1, len(lights)))(*lights)
        self.surfaces = (A.Surface * len(surfaces))(*surfaces)
        self.scene = A.Scene(self.shapes, len(shapes), self.materials, len(materials), self.lights, len(lights), self.surfaces,
                             len(surfaces), environment_light, camera)
        self.flat = C.pointer(self.scene)


def make_shape(A, kind, pts=(), normal=None, radius=0.0, flip=False):
    s = A.Shape()
    s.kind = kind
    for i, p in enumerate(pts):
        for j in range(3):
            s.p[i][j] = float(p[j])
    if normal is None and kind in (A.SHAPE_TRIANGLE, A.SHAPE_RECTANGLE):
        p0, p1, p2 = (np.array(pts[i], np.float32) for i in range(3))
        n = np.cross(p1 - p0, p2 - p0).astype(np.float32)
        normal = n / np.float32(np.sqrt(np.float32((n * n).sum())))
        if flip:
            normal = -normal
    if normal is not None:
        for j in range(3):
            s.normal[j] = float(normal[j])
    s.radius = radius
    return s


def make_material(A, kind, c0=(0, 0, 0), c1=(0, 0, 0), eta=0.0, exponent=0.0):
    m = A.Material()
    m.kind = kind
    for j in range(3):
        m.color0[j] = c0[j]
        m.color1[j] = c1[j]
    m.eta, m.exponent = eta, exponent
    if kind == A.MATERIAL_PLASTIC:
        lum = lambda c: np.float32(0.212671) * np.float32(c[0]) + np.float32(0.715160) * np.float32(c[1]) + np.float32(0.072169) * np.float32(c[2])
        d, sp = lum(c0), lum(c1)
        m.diffuse_probability, m.specular_probability = float(d / (d + sp)), float(sp / (d + sp))
    return m


def make_light(A, kind, color, shape=-1, position=(0, 0, 0), direction=(0, 0, -1), world_radius=0.0):
    l = A.Light()
    l.kind, l.shape = kind, shape
    for j in range(3):
        l.color[j], l.position[j], l.direction[j] = color[j], position[j], direction[j]
    l.world_radius = world_radius
    return l
