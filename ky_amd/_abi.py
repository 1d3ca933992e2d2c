"""ctypes mirror of include/kyhip.h (struct layouts, enums) and loaders for the in-tree shared libraries.

Plumbing only.  The product library is ky_amd/lib/libkyhip.so (HIP kernels + C ABI); if it is missing the
import of anything that needs it fails loudly -- there is no CPU fallback in this package.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(_HERE, "lib")
REPO_ROOT = os.path.dirname(_HERE)

# ---- enums (include/kyhip.h) ----
SHAPE_DISK, SHAPE_TRIANGLE, SHAPE_RECTANGLE, SHAPE_SPHERE = 0, 1, 2, 3
MATERIAL_MATTE, MATERIAL_MIRROR, MATERIAL_GLASS, MATERIAL_PLASTIC = 0, 1, 2, 3
LIGHT_POINT, LIGHT_DIRECTION, LIGHT_AREA, LIGHT_ENVIRONMENT = 0, 1, 2, 3
INTEGRATOR_POSITION, INTEGRATOR_NORMAL, INTEGRATOR_BASECOLOR = 0, 1, 2
INTEGRATOR_DIRECT_LIGHTING, INTEGRATOR_PATH_TRACING_ITERATION = 6, 11
INTEGRATOR_SIMPLE_PATH_TRACING_RECURSION, INTEGRATOR_PATH_TRACING_RECURSION, INTEGRATOR_PATH_TRACING_RECURSION_DEFERED = 8, 9, 10
DIRECT_IDLE, DIRECT_BSDF, DIRECT_LIGHT, DIRECT_BSDF_MIS, DIRECT_LIGHT_MIS, DIRECT_BOTH_MIS = 0, 4, 8, 16, 32, 48
SAMPLER_DEBUG, SAMPLER_RANDOM = 0, 1
SP_VARIANT_SMALLPT, SP_VARIANT_REWRITE = 0, 1
KY_OK, KY_ERR_INVALID_VALUE, KY_ERR_LIMIT, KY_ERR_DEVICE, KY_ERR_NO_DEVICE = 0, -1, -2, -3, -4
# hard limits of the device path (include/kyhip.h)
MAX_SHAPES, MAX_SURFACES, MAX_MATERIALS, MAX_LIGHTS = 256, 256, 64, 16

# cornell_box_enum_t (ky.cpp:3121-3145)
CB_LIGHT_AREA, CB_LIGHT_DIRECTION, CB_LIGHT_POINT, CB_LIGHT_ENVIRONMENT = 1, 2, 4, 8
CB_LARGE_MIRROR, CB_LARGE_GLASS, CB_SMALL_MIRROR, CB_SMALL_GLASS = 16, 32, 64, 128
CB_BOTH_SMALL_SPHERES = CB_SMALL_MIRROR | CB_SMALL_GLASS
CB_DEFAULT_SCENE = CB_BOTH_SMALL_SPHERES | CB_LIGHT_AREA

f3 = C.c_float * 3


class Shape(C.Structure):
    _fields_ = [("kind", C.c_int32), ("p", (C.c_float * 3) * 4), ("normal", f3), ("radius", C.c_float)]


class Material(C.Structure):
    _fields_ = [("kind", C.c_int32), ("color0", f3), ("color1", f3), ("eta", C.c_float), ("exponent", C.c_float),
                ("diffuse_probability", C.c_float), ("specular_probability", C.c_float)]


class Light(C.Structure):
    _fields_ = [("kind", C.c_int32), ("shape", C.c_int32), ("color", f3), ("position", f3), ("direction", f3),
                ("world_radius", C.c_float)]


class Surface(C.Structure):
    _fields_ = [("shape", C.c_int32), ("material", C.c_int32), ("area_light", C.c_int32)]


class Camera(C.Structure):
    _fields_ = [("position", f3), ("front", f3), ("right", f3), ("up", f3), ("resolution", C.c_float * 2)]


class Scene(C.Structure):
    _fields_ = [("shapes", C.POINTER(Shape)), ("shape_count", C.c_int32),
                ("materials", C.POINTER(Material)), ("material_count", C.c_int32),
                ("lights", C.POINTER(Light)), ("light_count", C.c_int32),
                ("surfaces", C.POINTER(Surface)), ("surface_count", C.c_int32),
                ("environment_light", C.c_int32), ("camera", Camera)]


class RenderParams(C.Structure):
    _fields_ = [("integrator", C.c_int32), ("max_path_depth", C.c_int32), ("direct_sample", C.c_int32),
                ("samples_per_pixel", C.c_int32), ("sampler", C.c_int32), ("seed", C.c_uint32),
                ("width", C.c_int32), ("height", C.c_int32), ("tile_w", C.c_int32), ("tile_h", C.c_int32),
                ("tile_first", C.c_int32), ("tile_step", C.c_int32)]


FP = C.POINTER(C.c_float)
SP = C.POINTER(Scene)
PP = C.POINTER(RenderParams)

# every symbol include/kyhip.h declares: name -> (restype, argtypes)
class SmallptSphere(C.Structure):   # ky_smallpt_sphere
    _fields_ = [("rad", C.c_double), ("p", C.c_double * 3), ("e", C.c_double * 3), ("c", C.c_double * 3), ("refl", C.c_int), ("pad_", C.c_int)]


class SmallptParams(C.Structure):   # ky_smallpt_params
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("samps", C.c_int), ("seed", C.c_uint32), ("max_depth", C.c_int),
                ("variant", C.c_int)]


SP_DIFF, SP_SPEC, SP_REFR = 0, 1, 2
SSP = C.POINTER(SmallptSphere)
SPP = C.POINTER(SmallptParams)

KYHIP_SYMBOLS = {
    "kyhip_last_error": (C.c_char_p, []),
    "kyhip_set_engine": (C.c_int, [C.c_int]),
    "kyhip_set_specialisation": (C.c_int, [C.c_int]),
    "kyhip_set_boxes": (C.c_int, [C.c_int]),
    "kyhip_set_shadow_queue": (C.c_int, [C.c_int]),
    "kyhip_set_jit": (C.c_int, [C.c_int]),
    "kyhip_jit_status": (C.c_char_p, []),
    "kyhip_jit_failures": (C.c_int, []),
    "kyhip_multi_status": (C.c_char_p, [C.c_int]),
    "kyhip_film_alloc": (C.c_void_p, [C.c_size_t]),
    "kyhip_film_free": (None, [C.c_void_p]),
    "kyhip_seam_threads": (C.c_int, []),
    "kyhip_jit_compile": (C.c_int64, [C.c_char_p]),
    "kyhip_kernel_source_hash": (C.c_uint64, []),
    "kyhip_abi_version": (C.c_int, []),
    "kyhip_device_count": (C.c_int, []),
    "kyhip_shard_tile_count": (C.c_int64, [PP]),
    "kyhip_shard_float_count": (C.c_int64, [PP]),
    "kyhip_render": (C.c_int, [C.c_int, SP, PP, C.c_void_p, C.c_size_t]),
    "kyhip_workspace_bytes": (C.c_size_t, [PP]),
    "kyhip_render_tiles_device": (C.c_int, [C.c_int, SP, PP, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "kyhip_film_add_tiles_device": (C.c_int, [C.c_int, PP, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "kyhip_film_add_gathered_device": (C.c_int, [C.c_int, PP, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]),
    "kyhip_render_multi": (C.c_int, [C.POINTER(C.c_int), C.c_int, SP, PP, C.c_void_p, C.c_size_t]),
    "kyhip_kernel_ms": (C.c_float, [C.c_int]),
    "kyhip_last_kernel": (C.c_char_p, [C.c_int]),
    "kyhip_kat_nee": (C.c_int, [C.c_int, SP, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "kyhip_kat_li_trace": (C.c_int, [C.c_int, SP, PP, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "kyhip_smallpt_scene": (C.c_int, [SSP]),
    "kyhip_smallpt_scene_rewrite": (C.c_int, [SSP]),
    "kyhip_smallpt_render": (C.c_int, [C.c_int, SSP, C.c_int, SPP, C.c_void_p]),
    "kyhip_smallpt_kat_radiance": (C.c_int, [C.c_int, SSP, C.c_int, SPP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "kyhip_kat_intersect": (C.c_int, [C.c_int, C.POINTER(Shape), C.c_void_p, C.c_int, C.c_void_p]),
    "kyhip_kat_camera": (C.c_int, [C.c_int, C.POINTER(Camera), C.c_void_p, C.c_int, C.c_void_p]),
    "kyhip_kat_bsdf": (C.c_int, [C.c_int, C.POINTER(Material), C.c_void_p, C.c_int, C.c_void_p]),
    "kyhip_kat_light": (C.c_int, [C.c_int, SP, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "kyhip_kat_scene_intersect": (C.c_int, [C.c_int, SP, C.c_void_p, C.c_int, C.c_void_p]),
    "kyhip_kat_occluded": (C.c_int, [C.c_int, SP, C.c_void_p, C.c_int, C.c_void_p]),
    "kyhip_kat_any_pair": (C.c_int, [C.c_int, SP, C.c_void_p, C.c_int, C.c_void_p]),
    "kyhip_kat_occluded_between": (C.c_int, [C.c_int, SP, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "kyhip_scene_non_occluders": (C.c_int, [SP, C.c_int, C.c_void_p, C.c_int]),
    "kyhip_scene_boxes": (C.c_int, [SP, C.c_void_p, C.c_int]),
    "kyhip_scene_facts": (C.c_int, [SP]),
    "kyhip_kat_li": (C.c_int, [C.c_int, SP, PP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
}

KYHOST_SYMBOLS = {
    "kyhost_last_error": (C.c_char_p, []),
    "kyhost_scene_create_cornell_box": (C.c_void_p, [C.c_int, C.c_float, C.c_float]),
    "kyhost_scene_create_mis": (C.c_void_p, [C.c_float, C.c_float]),
    "kyhost_scene_destroy": (None, [C.c_void_p]),
    "kyhost_scene_flatten": (SP, [C.c_void_p]),
    "kyhost_render": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint, C.c_int, C.c_int,
                                C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]),
    "kyhost_debug_area": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint, C.c_int, C.c_int,
                                    C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "kyhost_store_image": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "kyhost_gamma_encoding": (C.c_int, [C.c_float]),
}


# KY_SANITIZE=asan (test infrastructure: `make sanitize`, tests/test_sanitize.py): load the sanitizer builds of the HOST-ONLY code under build/san
# instead of the product libraries -- ky_pack.cpp / ky_jit.cpp as libkyhip_host_asan.so (no kernels, no HIP runtime: every entry point that needs a
# GPU is simply absent and raises AttributeError when called), the host mirror as libkyhost_asan.so.  The process must have been started with the
# sanitizer runtime preloaded (tools/sanitize/run.sh).  Never set in production: the product libraries are the only ones that render.
SANITIZE = os.environ.get("KY_SANITIZE")
SAN_DIR = os.path.join(REPO_ROOT, "build", "san")

KYHOSTCHECK_SYMBOLS = {   # ky_amd/csrc/ky_hostcheck.cpp: present in the sanitizer builds only
    "kyhostcheck_pack": (C.c_int, [SP, C.POINTER(C.c_int), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "kyhostcheck_chunks": (C.c_int, [C.c_int]),
    "kyhostcheck_shard": (C.c_longlong, [PP]),
    "kyhostcheck_add_rows": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "kyhostcheck_seam_stress": (C.c_int, [C.c_int]),
    "kyhostcheck_jit_stress": (C.c_int, [C.c_int, C.c_int]),
}


def _bind(lib, table, partial=False):
    for name, (res, args) in table.items():
        try:
            fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        except AttributeError:
            if partial:
                continue
            raise
        fn.restype = res
        fn.argtypes = args
    return lib


_kyhip = None
_kyhost = None
_hiprt = None


def load_hip_runtime():
    """Put exactly ONE HIP runtime into the process's global symbol scope before libkyhip.so is loaded.

    libkyhip.so is linked with -no-hip-rt.  The PyTorch wheel bundles its own libamdhip64.so (no SONAME), so a
    library that pinned /opt/rocm/lib/libamdhip64.so.7 would bring a SECOND runtime into the process and torch's
    streams / device pointers would be foreign to it.  When torch is installed its runtime is the one we share;
    otherwise (or with KYHIP_HIP_RUNTIME set) the system runtime is used.
    """
    global _hiprt
    if _hiprt is not None:
        return _hiprt
    candidates = []
    if os.environ.get("KYHIP_HIP_RUNTIME"):
        candidates.append(os.environ["KYHIP_HIP_RUNTIME"])
    else:
        try:
            import torch  # noqa: F401  (loads torch/lib/libamdhip64.so into the process)
            candidates.append(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
        except Exception:
            pass
        candidates += ["/opt/rocm/lib/libamdhip64.so", "libamdhip64.so"]
    err = None
    for path in candidates:
        if os.path.isabs(path) and not os.path.exists(path):
            continue
        try:
            _hiprt = C.CDLL(path, mode=C.RTLD_GLOBAL)
            return _hiprt
        except OSError as e:  # try the next candidate
            err = e
    raise ImportError(f"no HIP runtime (libamdhip64.so) could be loaded: {err}")


def load_kyhip():
    """Load the product library.  Raises if it has not been built: no fallback."""
    global _kyhip
    if _kyhip is None and SANITIZE:
        path = os.path.join(SAN_DIR, "libkyhip_host_%s.so" % SANITIZE)
        if not os.path.exists(path):
            raise ImportError(f"{path} is missing: `make sanitize-build`")
        _kyhip = _bind(_bind(C.CDLL(path, mode=C.RTLD_GLOBAL), KYHIP_SYMBOLS, partial=True), KYHOSTCHECK_SYMBOLS)
    if _kyhip is None:
        path = os.environ.get("KYHIP_LIB") or os.path.join(LIB_DIR, "libkyhip.so")  # KYHIP_LIB: A/B builds of the same ABI
        if not os.path.exists(path):
            raise ImportError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(ky_amd has no CPU fallback)")
        load_hip_runtime()
        _kyhip = _bind(C.CDLL(path, mode=C.RTLD_GLOBAL), KYHIP_SYMBOLS)
    return _kyhip


def load_kyhost():
    global _kyhost
    if _kyhost is None:
        load_kyhip()
        path = os.path.join(SAN_DIR, "libkyhost_%s.so" % SANITIZE) if SANITIZE else os.path.join(LIB_DIR, "libkyhost.so")
        if not os.path.exists(path):
            raise ImportError(f"{path} is missing: build it with `make` at the repo root")
        _kyhost = _bind(C.CDLL(path), KYHOST_SYMBOLS)
    return _kyhost
