"""Python plumbing over the C ABI (include/kyhip.h) and the C++ host layer (ky_amd/host/ky.hpp).

Nothing here computes radiance: every function forwards to libkyhip.so (HIP kernels) or to the C++ host
classes.  numpy is used for host buffers only.
"""
import ctypes as C

import numpy as np

from . import _abi as A


class KyError(RuntimeError):
    pass


def _check(rc, lib=None):
    if rc != A.KY_OK:
        lib = lib or A.load_kyhip()
        raise KyError(f"kyhip error {rc}: {lib.kyhip_last_error().decode()}")


def _fptr(a):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


class SceneHandle:
    """A scene built by the C++ host layer (scene_t::create_*_scene) plus its flat C-ABI view."""

    def __init__(self, ptr):
        if not ptr:
            raise KyError("scene creation failed: " + A.load_kyhost().kyhost_last_error().decode())
        self._ptr = C.c_void_p(ptr)
        self._host = A.load_kyhost()
        self.flat = self._host.kyhost_scene_flatten(self._ptr)
        if not self.flat:
            raise KyError("scene flatten failed: " + self._host.kyhost_last_error().decode())

    def __del__(self):
        try:
            if self._ptr:
                self._host.kyhost_scene_destroy(self._ptr)
                self._ptr = None
        except Exception:
            pass

    @property
    def c(self):
        return self.flat.contents

    @property
    def ptr(self):
        return self._ptr


def cornell_box_scene(flags, width, height):
    """scene_t::create_cornell_box_scene(flags, {width, height}) -- ky.cpp:3240."""
    return SceneHandle(A.load_kyhost().kyhost_scene_create_cornell_box(int(flags), float(width), float(height)))


def mis_scene(width, height):
    """scene_t::create_mis_scene({width, height}) -- ky.cpp:3434."""
    return SceneHandle(A.load_kyhost().kyhost_scene_create_mis(float(width), float(height)))


def make_params(width, height, spp, integrator=A.INTEGRATOR_PATH_TRACING_ITERATION, max_path_depth=5,
                direct_sample=A.DIRECT_BOTH_MIS, sampler=A.SAMPLER_RANDOM, seed=1234, tile_w=16, tile_h=16,
                tile_first=0, tile_step=1):
    return A.RenderParams(integrator, max_path_depth, direct_sample, spp, sampler, seed, width, height, tile_w, tile_h,
                          tile_first, tile_step)


def _scene_ptr(scene):
    """A SceneHandle, any object with a `.flat` POINTER(Scene) (e.g. a scene assembled from ctypes structs), or the pointer itself."""
    return scene.flat if hasattr(scene, "flat") else scene


def render(scene, params, film=None, device=0, row_stride_px=None, origin_px=(0, 0)):
    """kyhip_render: integrator_t::render(scene, sampler, film) on the GPU; returns the (accumulated) host film."""
    lib = A.load_kyhip()
    if film is None:
        film = np.zeros((params.height, params.width, 3), np.float32)
    stride = film.shape[1] if row_stride_px is None else row_stride_px
    base = film.ctypes.data + (origin_px[1] * stride + origin_px[0]) * 12
    _check(lib.kyhip_render(device, _scene_ptr(scene), C.byref(params), C.c_void_p(base), stride))
    return film


class PinnedFilm:
    """A [height, width, 3] float32 numpy view of pinned host memory from kyhip_film_alloc (include/kyhip.h): a film the GPU adds to in place.
    `array` stays valid while this object lives."""

    def __init__(self, height, width):
        lib = A.load_kyhip()
        self._lib, self.nbytes = lib, height * width * 12
        self._ptr = lib.kyhip_film_alloc(self.nbytes)
        if not self._ptr:
            raise MemoryError("kyhip_film_alloc returned NULL (no device?)")
        self.array = np.ctypeslib.as_array((C.c_float * (height * width * 3)).from_address(self._ptr)).reshape(height, width, 3)
        self.array[...] = 0

    def __del__(self):
        if getattr(self, "_ptr", None):
            self.array = None
            self._lib.kyhip_film_free(self._ptr)
            self._ptr = None


def render_multi(scene, params, devices, film=None, row_stride_px=None, origin_px=(0, 0)):
    """kyhip_render_multi: integrator_t::render with the frame's tiles spread over the listed GPUs (a device may repeat)."""
    lib = A.load_kyhip()
    if film is None:
        film = np.zeros((params.height, params.width, 3), np.float32)
    stride = film.shape[1] if row_stride_px is None else row_stride_px
    base = film.ctypes.data + (origin_px[1] * stride + origin_px[0]) * 12
    devs = (C.c_int * len(devices))(*devices)
    _check(lib.kyhip_render_multi(devs, len(devices), _scene_ptr(scene), C.byref(params), C.c_void_p(base), stride))
    return film


def render_host_api(scene, integrator_enum, depth, direct_sample, sampler, spp, width, height, seed=1234,
                    grid=None, cell=0, film=None, device=0):
    """Drive the C++ host classes exactly like a reference driver: create_integrator(...)->render(&scene, sampler, &film)."""
    host = A.load_kyhost()
    rows, cols = grid if grid else (0, 0)
    fw, fh = (cols * width, rows * height) if grid else (width, height)
    if film is None:
        film = np.zeros((fh, fw, 3), np.float32)
    rc = host.kyhost_render(scene.ptr, integrator_enum, depth, direct_sample, sampler, spp, seed, width, height, rows, cols,
                            cell, _fptr(film), device)
    if rc == -2:
        return None  # create_integrator returned nullptr (ky.cpp:4638)
    if rc != 0:
        raise KyError("kyhost_render failed: " + host.kyhost_last_error().decode())
    return film


def debug_area_host_api(scene, integrator_enum, depth, direct_sample, sampler, spp, width, height, begin, end, film=None, seed=1234, device=0):
    """create_integrator(...)->debug_area(&scene, sampler, &film, begin, end) (ky.cpp:3733-3777) through the C++ host classes; a 1 x 1 area
    goes through debug_pixel (3784).  Returns the film (modified in place when given)."""
    host = A.load_kyhost()
    if film is None:
        film = np.zeros((height, width, 3), np.float32)
    rc = host.kyhost_debug_area(scene.ptr, integrator_enum, depth, direct_sample, sampler, spp, seed, width, height, _fptr(film),
                                int(begin[0]), int(begin[1]), int(end[0]), int(end[1]), device)
    if rc == -2:
        return None
    if rc != 0:
        raise KyError("kyhost_debug_area failed: " + host.kyhost_last_error().decode())
    return film


def kernel_ms(device=0):
    return float(A.load_kyhip().kyhip_kernel_ms(device))


# ---- function-level entry points (parity tests) -------------------------------------------------

def kat_intersect(shape, rays7, device=0):
    lib = A.load_kyhip()
    rays7 = np.ascontiguousarray(rays7, np.float32)
    out = np.zeros((rays7.shape[0], 8), np.float32)
    _check(lib.kyhip_kat_intersect(device, C.byref(shape), _fptr(rays7), rays7.shape[0], _fptr(out)))
    return out


def kat_camera(camera, p_film2, device=0):
    lib = A.load_kyhip()
    p_film2 = np.ascontiguousarray(p_film2, np.float32)
    out = np.zeros((p_film2.shape[0], 6), np.float32)
    _check(lib.kyhip_kat_camera(device, C.byref(camera), _fptr(p_film2), p_film2.shape[0], _fptr(out)))
    return out


def kat_bsdf(material, in12, device=0):
    lib = A.load_kyhip()
    in12 = np.ascontiguousarray(in12, np.float32)
    out = np.zeros((in12.shape[0], 13), np.float32)
    _check(lib.kyhip_kat_bsdf(device, C.byref(material), _fptr(in12), in12.shape[0], _fptr(out)))
    return out


def kat_light(scene, light, in11, device=0):
    lib = A.load_kyhip()
    in11 = np.ascontiguousarray(in11, np.float32)
    out = np.zeros((in11.shape[0], 11), np.float32)
    _check(lib.kyhip_kat_light(device, _scene_ptr(scene), light, _fptr(in11), in11.shape[0], _fptr(out)))
    return out


def kat_scene_intersect(scene, rays7, device=0):
    lib = A.load_kyhip()
    rays7 = np.ascontiguousarray(rays7, np.float32)
    out = np.zeros((rays7.shape[0], 9), np.float32)
    _check(lib.kyhip_kat_scene_intersect(device, _scene_ptr(scene), _fptr(rays7), rays7.shape[0], _fptr(out)))
    return out


def kat_any_pair(scene, in13, device=0):
    """trace_any_pair (the environment estimate's scan): n x {o_a, d_a, o_b, d_b, tmax_b} -> n x {A meets a surface, B meets one before tmax_b}."""
    lib = A.load_kyhip()
    in13 = np.ascontiguousarray(in13, np.float32)
    out = np.zeros((in13.shape[0], 2), np.float32)
    _check(lib.kyhip_kat_any_pair(device, _scene_ptr(scene), _fptr(in13), in13.shape[0], _fptr(out)))
    return out


def kat_occluded(scene, in9, device=0, table=None):
    """scene_t::occluded for n x {p, normal, target}.  table None: every surface is tested; -1 / a light index: the occluder table the
    render kernels use for segments between scene points / for shadow rays towards samples of that light (kyhip_kat_occluded_between)."""
    lib = A.load_kyhip()
    in9 = np.ascontiguousarray(in9, np.float32)
    out = np.zeros((in9.shape[0],), np.float32)
    if table is None:
        _check(lib.kyhip_kat_occluded(device, _scene_ptr(scene), _fptr(in9), in9.shape[0], _fptr(out)))
    else:
        _check(lib.kyhip_kat_occluded_between(device, _scene_ptr(scene), int(table), _fptr(in9), in9.shape[0], _fptr(out)))
    return out


def scene_non_occluders(scene, light=-1):
    """kyhip_scene_non_occluders (host only), per surface: 1 not in the occluder table for `light` (-1: rays that end on a scene point),
    2 scanned only for rays with an end behind that light's plane, 0 always tested."""
    lib = A.load_kyhip()
    n = _scene_ptr(scene).contents.surface_count
    out = np.zeros(max(1, n), np.int32)
    rc = lib.kyhip_scene_non_occluders(_scene_ptr(scene), int(light), out.ctypes.data_as(C.c_void_p), n)
    if rc < 0:
        _check(rc)
    return out[:n].copy()


def scene_facts(scene):
    """kyhip_scene_facts (host only): the KY_FEAT_* mask the library finds for the scene."""
    lib = A.load_kyhip()
    rc = lib.kyhip_scene_facts(_scene_ptr(scene))
    if rc < 0:
        _check(rc)
    return rc


def scene_boxes(scene):
    """kyhip_scene_boxes (host only): (number of boxes, per surface 8 * box + 2 * axis + side for a surface that is a whole face of an axis-aligned box
    the nearest-hit traversal tests with one slab test, -1 otherwise)."""
    lib = A.load_kyhip()
    n = _scene_ptr(scene).contents.surface_count
    out = np.zeros(max(1, n), np.int32)
    rc = lib.kyhip_scene_boxes(_scene_ptr(scene), out.ctypes.data_as(C.c_void_p), n)
    if rc < 0:
        _check(rc)
    return rc, out[:n].copy()


def kat_li(scene, params, x, y, s0, n, device=0):
    lib = A.load_kyhip()
    out = np.zeros((n, 3), np.float32)
    _check(lib.kyhip_kat_li(device, _scene_ptr(scene), C.byref(params), x, y, s0, n, _fptr(out)))
    return out


def kat_nee(scene, direct_sample, light, in15, device=0):
    lib = A.load_kyhip()
    in15 = np.ascontiguousarray(in15, np.float32)
    out = np.zeros((in15.shape[0], 6), np.float32)
    _check(lib.kyhip_kat_nee(device, _scene_ptr(scene), direct_sample, light, _fptr(in15), in15.shape[0], _fptr(out)))
    return out


def kat_li_trace(scene, params, x, y, s, max_rows=64, device=0):
    """kyhip_kat_li_trace: (rows [n, 26], li [3]) of one camera sample of path_tracing_iteration_t."""
    lib = A.load_kyhip()
    rows = np.zeros((max_rows, 26), np.float32)
    li = np.zeros(3, np.float32)
    n = lib.kyhip_kat_li_trace(device, _scene_ptr(scene), C.byref(params), x, y, s, _fptr(rows), max_rows, _fptr(li))
    if n < 0:
        _check(n, lib)
    return rows[:n], li


# ---- smallpt's own scene in double precision --------------------------------------

def smallpt_scene():
    """The 9 spheres of smallpt2pbrt/smallpt.cpp:42-52 as a ctypes array."""
    spheres = (A.SmallptSphere * 9)()
    n = A.load_kyhip().kyhip_smallpt_scene(spheres)
    assert n == 9
    return spheres


def smallpt_scene_rewrite():
    """Scene::CreateSmallptScene of smallpt2pbrt/smallpt_rewrite.cpp:1199-1244 (the same spheres at -z)."""
    spheres = (A.SmallptSphere * 9)()
    n = A.load_kyhip().kyhip_smallpt_scene_rewrite(spheres)
    assert n == 9
    return spheres


def smallpt_params(width, height, samps, seed=1234, max_depth=10, variant=A.SP_VARIANT_SMALLPT):
    return A.SmallptParams(width, height, samps, seed, max_depth, variant)


def smallpt_render(spheres, params, device=0):
    """kyhip_smallpt_render: smallpt's main() loop nest on the GPU; returns float64 [H, W, 3], row 0 = top of the picture."""
    lib = A.load_kyhip()
    img = np.zeros((params.height, params.width, 3), np.float64)
    _check(lib.kyhip_smallpt_render(device, spheres, len(spheres), C.byref(params), img.ctypes.data_as(C.c_void_p)), lib)
    return img


def smallpt_kat_radiance(spheres, params, x, y, sx, sy, s0, n, device=0):
    lib = A.load_kyhip()
    out = np.zeros((n, 3), np.float64)
    _check(lib.kyhip_smallpt_kat_radiance(device, spheres, len(spheres), C.byref(params), x, y, sx, sy, s0, n,
                                          out.ctypes.data_as(C.c_void_p)), lib)
    return out


def store_image(filename, rgb, kind="bmp"):
    """film_t::store_{ppm,bmp,hdr}_impl -- ky.cpp:1646-1782."""
    host = A.load_kyhost()
    rgb = np.ascontiguousarray(rgb, np.float32)
    k = {"ppm": 0, "bmp": 1, "hdr": 2}[kind]
    if host.kyhost_store_image(filename.encode(), k, rgb.shape[1], rgb.shape[0], _fptr(rgb)) != 0:
        raise KyError("store_image failed: " + host.kyhost_last_error().decode())
