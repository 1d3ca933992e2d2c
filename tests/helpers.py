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


def prove_ties(rows, gpu_answer, oracle_answer, perturb, rng, what, tries=64):
    """Every discrete disagreement between the HIP path and the oracle must be a TIE: the oracle itself gives the GPU's answer when the input is moved by a few
    1e-5 of the scene's size (an edge, a silhouette, a hit at epsilon or tmax, two surfaces at one distance) -- nothing else may differ.
    rows: the disagreeing inputs; gpu_answer(i) -> what the GPU said for row i (any comparable value); oracle_answer(row) -> the oracle's answer for one input
    row; perturb(row, rng) -> a slightly moved copy."""
    for i, row in rows:
        want = gpu_answer(i)
        for _ in range(tries):
            if oracle_answer(perturb(row, rng)) == want:
                break
        else:
            raise AssertionError((what + ": the two sides differ away from any threshold", i, row, want, oracle_answer(row)))


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


# ---- vertex traces (kyhip_kat_li_trace / kyo_trace_li rows of 26 floats): what makes two traces of one camera sample differ ----
T_GEOM = slice(3, 12)      # position, normal, wo
T_BETA = slice(12, 15)
T_LO = slice(15, 18)
T_DECISIONS = (1, 2, 23, 24, 25)   # surface, lobe, sampled lobe flags, which BSDF-sampling / light-sampling estimates were non-black


def trace_close(a, b, tol):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return bool(np.all(np.abs(a - b) <= tol * np.maximum(1.0, np.maximum(np.abs(a), np.abs(b)))))


def explain_sample(g_rows, c_rows, value_tol=2e-4, geom_tol=1e-4, amplifying_surfaces=()):
    """Why do the two traces of one camera sample differ?  -> "decision" (a recorded discrete decision differs first: nearest hit, lobe,
    glass branch, an estimate's zero / non-zero outcome, termination), "specular" (the first continuous difference lies at or after a
    vertex on a mirror / glass surface: a curved specular surface amplifies the rounding of a grazing hit 10-50x), "phong" (at or
    after a vertex whose lobe is the Phong lobe: pow amplifies the rounding of its base by the exponent), or None: a continuous
    difference with nothing to amplify it -- which no test accepts.  value_tol: how far throughput and radiance may drift with equal
    decisions before it counts as a difference (2e-4; scenes with small sphere lights need more: uniform_cone_pdf's 1 - cos(theta_max)
    (ky.cpp:798, 1510-1512) cancels to 1e-3 of its terms for a light that subtends 1e-4 sr, in the reference's arithmetic as in any other);
    a sample that stays inside it everywhere is "within tolerance".  geom_tol: the same for position / normal / wo (1e-4; the normal of a
    sphere of radius r carries 1 / r times the rounding of the hit point, which sphere_t::intersect's cancelling discriminant leaves at
    ~1e-5 for a small sphere seen from afar).  amplifying_surfaces: caller's surface indices that amplify rounding like a specular sphere does
    whatever their material -- spheres a few hundredths across, whose grazing hits move by 1e-3 and whose normals turn by 1e-2 on an ulp of the
    ray: a first continuous difference at or after a vertex on one of them is "small sphere"."""
    specular = phong = small = grazing = False
    for k in range(min(len(g_rows), len(c_rows))):
        g, c = g_rows[k], c_rows[k]
        if any(g[j] != c[j] for j in T_DECISIONS):
            return "decision"
        lobe = int(c[2])
        specular = specular or lobe in (1, 2)   # "at or after": the hit ON a small specular sphere is where the cancellation happens
        small = small or int(c[1]) in amplifying_surfaces
        # a ray that meets its surface at a grazing angle turns an offset d of its origin into d / cos of the hit point (round 6: a ceiling vertex 1.8 mm from the back
        # wall, its 1.4e-6 of rounding 4.5e-5 at the wall under cos = 0.03, and 6e-4 three bounces on): at or after such a hit a continuous difference is "grazing"
        grazing = grazing or abs(float(np.dot(np.asarray(c[6:9], np.float64), np.asarray(c[9:12], np.float64)))) < 0.05
        same = trace_close(g[T_GEOM], c[T_GEOM], geom_tol) and trace_close(g[T_BETA], c[T_BETA], value_tol)
        if same:   # this vertex's own radiance so far: continuous in equal inputs, but a Phong value here is already amplified
            same = trace_close(g[T_LO], c[T_LO], value_tol)
            if not same and lobe == 3:
                phong = True
        if not same:
            return "specular" if specular else ("phong" if phong else ("small sphere" if small else ("grazing" if grazing else None)))
        phong = phong or lobe == 3
    if len(g_rows) != len(c_rows):
        return "decision"   # one path went on: roulette, the depth cap or the last traversal decided differently
    if value_tol > 2e-4 or geom_tol > 1e-4:
        return "within tolerance"
    return "specular" if specular else ("phong" if phong else "decision")   # equal vertices: the final traversal (hit / miss, emission side) differs


def explain_pixel(api, O, scene, params, x, y, value_tol=2e-4, geom_tol=1e-4, amplifying_surfaces=()):
    """All differing samples of one pixel, classified (explain_sample); asserts that none is unexplained.  -> {kind: count}"""
    n = params.samples_per_pixel
    g, c = api.kat_li(scene, params, x, y, 0, n), O.li(scene, params, x, y, 0, n)
    fin = np.isfinite(c).all(1)
    d = np.abs(g - c).max(axis=1)
    sc = np.maximum(1e-3, np.abs(c).max(axis=1))
    kinds = {}
    for s in np.flatnonzero(fin & (d / sc > 1e-3)):
        g_rows, _ = api.kat_li_trace(scene, params, x, y, int(s))
        c_rows = O.trace_li(scene, params, x, y, int(s))
        kind = explain_sample(g_rows, c_rows, value_tol, geom_tol, amplifying_surfaces)
        assert kind is not None, ("sample differs continuously with nothing that amplifies rounding", x, y, int(s), g[s], c[s])
        kinds[kind] = kinds.get(kind, 0) + 1
    return kinds


def rmse_with_explained_flips(api, O, scene, params, g, c, max_exempt=8, threshold=5e-3, value_tol=2e-4, geom_tol=1e-4, amplifying_surfaces=()):
    """RMSE(g, c) over the finite pixels WITHOUT the (at most max_exempt) pixels that are off by more than `threshold` -- each of which
    must be explained sample by sample (explain_pixel: a discrete decision differs first, or the difference starts at a specular / Phong
    vertex).  -> (rmse without them, rmse with them, number exempted)"""
    fin = np.isfinite(c).all(axis=2)
    d = np.where(fin[..., None], g.astype(np.float64) - c.astype(np.float64), 0.0)
    m = np.abs(d).max(axis=2)
    keep = fin.copy()
    ys, xs = np.nonzero(m > threshold)
    order = np.argsort(-m[ys, xs])[:max_exempt]
    for i in order:
        kinds = explain_pixel(api, O, scene, params, int(xs[i]), int(ys[i]), value_tol, geom_tol, amplifying_surfaces)
        assert sum(v for k, v in kinds.items() if k != "within tolerance") > 0, ("pixel off by %.2e without a differing sample" % m[ys[i], xs[i]], int(xs[i]), int(ys[i]))
        keep[ys[i], xs[i]] = False
    return float(np.sqrt(np.mean(d[keep] ** 2))), float(np.sqrt(np.mean(d[fin] ** 2))), int(len(order))
