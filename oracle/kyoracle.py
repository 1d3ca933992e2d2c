"""ctypes binding of oracle/libkyoracle.so (the CPU checker).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
Nothing under ky_amd/ imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from ky_amd import _abi as A

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None

COUNTER_NAMES = ["camera_samples", "traversals", "shadow_rays", "primitive_tests", "nee_vertices", "light_estimates",
                 "bsdf_path_samples", "path_iterations", "mis_bsdf_rays", "rr_draws", "shadow_occluded"]


def build():
    subprocess.check_call(["make", "-C", _HERE, "libkyoracle.so"], stdout=subprocess.DEVNULL)


def load():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "libkyoracle.so")
        if A.SANITIZE:   # `make sanitize`: the address + undefined-behaviour build of the same sources (build/san)
            path = os.path.join(A.SAN_DIR, "libkyoracle_%s.so" % A.SANITIZE)
        elif not os.path.exists(path):
            build()
        _lib = C.CDLL(path)
        _lib.kyo_render.restype = C.c_int
        _lib.kyo_render.argtypes = [A.SP, A.PP, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        _lib.kyo_li.argtypes = [A.SP, A.PP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        _lib.kyo_kat_intersect.argtypes = [C.POINTER(A.Shape), C.c_void_p, C.c_int, C.c_void_p]
        _lib.kyo_kat_camera.argtypes = [C.POINTER(A.Camera), C.c_void_p, C.c_int, C.c_void_p]
        _lib.kyo_kat_frame.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        _lib.kyo_kat_bsdf.argtypes = [C.POINTER(A.Material), C.c_void_p, C.c_int, C.c_void_p]
        _lib.kyo_kat_light.argtypes = [A.SP, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        _lib.kyo_kat_scene_intersect.argtypes = [A.SP, C.c_void_p, C.c_int, C.c_void_p]
        _lib.kyo_kat_occluded.argtypes = [A.SP, C.c_void_p, C.c_int, C.c_void_p]
        _lib.kyo_world_bounding_sphere.argtypes = [A.SP, C.c_void_p]
        _lib.kyo_max_threads.restype = C.c_int
        _lib.kyo_smallpt_render.argtypes = [A.SSP, C.c_int, A.SPP, C.c_int, C.c_void_p]
        _lib.kyo_smallpt_radiance.argtypes = [A.SSP, C.c_int, A.SPP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        _lib.kyo_smallpt_scene.argtypes = [A.SSP]
        _lib.kyo_sprw_render.argtypes = [A.SSP, C.c_int, A.SPP, C.c_int, C.c_void_p]
        _lib.kyo_sprw_radiance.argtypes = [A.SSP, C.c_int, A.SPP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        _lib.kyo_sprw_scene.argtypes = [A.SSP]
    return _lib


def _sp(scene):
    return scene.flat if hasattr(scene, "flat") else scene


def _f(a):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


def cpus_granted():
    """CPUs this process may really use: the affinity mask cut down to the cgroup's CPU quota.  OpenMP sizes its team by the mask alone; on a box that shows
    256 CPUs and grants two, a 256-thread team spins against the quota and a one-second render takes minutes."""
    import os
    granted = len(os.sched_getaffinity(0))
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            fields = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = fields[0], fields[1]
            else:
                quota, period = fields[0], open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().split()[0]
            if quota not in ("max", "-1"):
                granted = min(granted, max(1, int(round(int(quota) / int(period)))))
            break
        except Exception:
            continue
    return granted


def render(scene, params, film=None, threads=0, counters=False):
    lib = load()
    if threads <= 0:
        threads = cpus_granted()
    if film is None:
        film = np.zeros((params.height, params.width, 3), np.float32)
    cnt = np.zeros(len(COUNTER_NAMES), np.uint64) if counters else None
    rc = lib.kyo_render(_sp(scene), C.byref(params), _f(film), film.shape[1], threads,
                        cnt.ctypes.data_as(C.c_void_p) if counters else None)
    if rc != 0:
        raise ValueError(f"kyo_render returned {rc}")
    if counters:
        return film, dict(zip(COUNTER_NAMES, (int(v) for v in cnt)))
    return film


def li(scene, params, x, y, s0, n):
    out = np.zeros((n, 3), np.float32)
    rc = load().kyo_li(_sp(scene), C.byref(params), x, y, s0, n, _f(out))
    if rc != 0:
        raise ValueError(f"kyo_li returned {rc}")
    return out


def trace_li(scene, params, x, y, s, max_rows=64):
    rows = np.zeros((max_rows, 26), np.float32)
    lib = load()
    lib.kyo_trace_li.argtypes = [A.SP, A.PP, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
    n = lib.kyo_trace_li(_sp(scene), C.byref(params), x, y, s, _f(rows), max_rows)
    return rows[:n]


def kat_intersect(shape, rays7):
    rays7 = np.ascontiguousarray(rays7, np.float32)
    out = np.zeros((rays7.shape[0], 8), np.float32)
    load().kyo_kat_intersect(C.byref(shape), _f(rays7), rays7.shape[0], _f(out))
    return out


def kat_frame(in6):
    """frame_t(normal) and its to_local / to_world of v: n x {normal[3], v[3]} -> n x {s[3], t[3], n[3], local[3], world[3]}"""
    in6 = np.ascontiguousarray(in6, np.float32)
    out = np.zeros((in6.shape[0], 15), np.float32)
    load().kyo_kat_frame(_f(in6), in6.shape[0], _f(out))
    return out


def kat_camera(camera, p_film2):
    p_film2 = np.ascontiguousarray(p_film2, np.float32)
    out = np.zeros((p_film2.shape[0], 6), np.float32)
    load().kyo_kat_camera(C.byref(camera), _f(p_film2), p_film2.shape[0], _f(out))
    return out


def kat_bsdf(material, in12):
    in12 = np.ascontiguousarray(in12, np.float32)
    out = np.zeros((in12.shape[0], 13), np.float32)
    load().kyo_kat_bsdf(C.byref(material), _f(in12), in12.shape[0], _f(out))
    return out


def kat_light(scene, light, in11):
    in11 = np.ascontiguousarray(in11, np.float32)
    out = np.zeros((in11.shape[0], 11), np.float32)
    load().kyo_kat_light(_sp(scene), light, _f(in11), in11.shape[0], _f(out))
    return out


def kat_nee(scene, direct_sample, light, in15):
    in15 = np.ascontiguousarray(in15, np.float32)
    out = np.zeros((in15.shape[0], 6), np.float32)
    lib = load()
    lib.kyo_kat_nee.argtypes = [A.SP, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    rc = lib.kyo_kat_nee(_sp(scene), direct_sample, light, _f(in15), in15.shape[0], _f(out))
    if rc != 0:
        raise ValueError(f"kyo_kat_nee returned {rc}")
    return out


def kat_scene_intersect(scene, rays7):
    rays7 = np.ascontiguousarray(rays7, np.float32)
    out = np.zeros((rays7.shape[0], 9), np.float32)
    load().kyo_kat_scene_intersect(_sp(scene), _f(rays7), rays7.shape[0], _f(out))
    return out


def kat_occluded(scene, in9):
    in9 = np.ascontiguousarray(in9, np.float32)
    out = np.zeros((in9.shape[0],), np.float32)
    load().kyo_kat_occluded(_sp(scene), _f(in9), in9.shape[0], _f(out))
    return out


def world_bounding_sphere(scene):
    out = np.zeros(4, np.float32)
    load().kyo_world_bounding_sphere(_sp(scene), _f(out))
    return out


def max_threads():
    return int(load().kyo_max_threads())


# ---- smallpt_oracle.cpp: smallpt2pbrt/smallpt.cpp restated (double precision) ----

def smallpt_scene():
    spheres = (A.SmallptSphere * 9)()
    assert load().kyo_smallpt_scene(spheres) == 9
    return spheres


def smallpt_render(spheres, params, rng_mode=0):
    """rng_mode 0: the HIP path's per-sample streams; 1: smallpt's own erand48 walk along each image row."""
    img = np.zeros((params.height, params.width, 3), np.float64)
    rc = load().kyo_smallpt_render(spheres, len(spheres), C.byref(params), rng_mode, img.ctypes.data_as(C.c_void_p))
    if rc != 0:
        raise ValueError(f"kyo_smallpt_render returned {rc}")
    return img


def smallpt_radiance(spheres, params, x, y, sx, sy, s0, n):
    out = np.zeros((n, 3), np.float64)
    rc = load().kyo_smallpt_radiance(spheres, len(spheres), C.byref(params), x, y, sx, sy, s0, n, out.ctypes.data_as(C.c_void_p))
    if rc != 0:
        raise ValueError(f"kyo_smallpt_radiance returned {rc}")
    return out


# ---- smallpt_rewrite_oracle.cpp: smallpt2pbrt/smallpt_rewrite.cpp restated (double precision) ----

def sprw_scene():
    spheres = (A.SmallptSphere * 9)()
    assert load().kyo_sprw_scene(spheres) == 9
    return spheres


def sprw_render(spheres, params, rng_mode=0):
    """Integrater::Render into a cleared film: float64 [H, W, 3], row 0 = top.  rng_mode 1: the reference's own
    std::mt19937_64 per image row (byte-exact with oracle/_ref/smallpt_rewrite); 0: the HIP path's per-sample streams."""
    film = np.zeros((params.height, params.width, 3), np.float64)
    rc = load().kyo_sprw_render(spheres, len(spheres), C.byref(params), rng_mode, film.ctypes.data_as(C.c_void_p))
    if rc != 0:
        raise ValueError(f"kyo_sprw_render returned {rc}")
    return film


def sprw_radiance(spheres, params, x, y, s0, n):
    out = np.zeros((n, 3), np.float64)
    rc = load().kyo_sprw_radiance(spheres, len(spheres), C.byref(params), x, y, s0, n, out.ctypes.data_as(C.c_void_p))
    if rc != 0:
        raise ValueError(f"kyo_sprw_radiance returned {rc}")
    return out


def sprw_gamma_bytes(film):
    """GammaEncoding (smallpt_rewrite.cpp:494) with the C library's pow, like the reference binary."""
    film = np.ascontiguousarray(film, np.float64)
    out = np.zeros(film.shape, np.uint8)
    lib = load()
    lib.kyo_sprw_gamma_bytes.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
    assert lib.kyo_sprw_gamma_bytes(film.ctypes.data_as(C.c_void_p), film.size, out.ctypes.data_as(C.c_void_p)) == 0
    return out
