"""CPU: the C-ABI library loads, exports every symbol include/kyhip.h declares, and the ctypes mirrors match the C layouts."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "kyhip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(kyhip_\w+)\s*\(", src)))


def test_header_declares_what_python_binds(A):
    assert header_functions() == sorted(A.KYHIP_SYMBOLS)


def test_library_exports_every_declared_symbol(A):
    lib = A.load_kyhip()
    for name in header_functions():
        assert hasattr(lib, name), name
    assert lib.kyhip_abi_version() == 4
    out = subprocess.check_output(["nm", "-D", "--defined-only", os.path.join(A.LIB_DIR, "libkyhip.so")], text=True)
    exported = set(re.findall(r" T (kyhip_\w+)", out))
    assert set(header_functions()) <= exported


def test_struct_layouts_match_the_c_header(A, tmp_path):
    names = ["ky_shape", "ky_material", "ky_light", "ky_surface", "ky_camera", "ky_scene", "ky_render_params"]
    prog = "#include <stdio.h>\n#include \"kyhip.h\"\nint main(void){" + "".join(
        f'printf("%zu\\n", sizeof({n}));' for n in names) + \
        'printf("%zu %zu %zu\\n", offsetof(ky_scene, environment_light), offsetof(ky_scene, camera), offsetof(ky_light, world_radius));return 0;}'
    c = tmp_path / "sz.c"
    c.write_text(prog)
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(c), "-o", str(exe)])
    lines = subprocess.check_output([str(exe)], text=True).split("\n")
    sizes = [int(x) for x in lines[:7]]
    py = [C.sizeof(t) for t in (A.Shape, A.Material, A.Light, A.Surface, A.Camera, A.Scene, A.RenderParams)]
    assert sizes == py
    offs = [int(x) for x in lines[7].split()]
    assert offs == [A.Scene.environment_light.offset, A.Scene.camera.offset, A.Light.world_radius.offset]


def test_shard_arithmetic_is_pure_host_code(A, api):
    lib = A.load_kyhip()
    p = api.make_params(100, 70, 4, tile_w=32, tile_h=32)  # 4 x 3 tiles, ragged edges
    assert lib.kyhip_shard_tile_count(C.byref(p)) == 12
    assert lib.kyhip_shard_float_count(C.byref(p)) == 12 * 32 * 32 * 3
    counts = []
    for r in range(5):
        q = api.make_params(100, 70, 4, tile_w=32, tile_h=32, tile_first=r, tile_step=5)
        counts.append(lib.kyhip_shard_tile_count(C.byref(q)))
    assert counts == [3, 3, 2, 2, 2] and sum(counts) == 12
    bad = api.make_params(100, 70, 4, tile_w=30)  # not a multiple of 8
    assert lib.kyhip_shard_tile_count(C.byref(bad)) == A.KY_ERR_INVALID_VALUE
    assert b"invalid" in lib.kyhip_last_error()


def test_invalid_enums_are_rejected_without_a_device(A, api):
    """create_integrator returns nullptr for unknown integrators (ky.cpp:4638); sample_all_light has no
    estimator for OR-ed / unknown strategy values (ky.cpp:3860)."""
    lib = A.load_kyhip()
    for bad in (api.make_params(8, 8, 1, integrator=7), api.make_params(8, 8, 1, integrator=12), api.make_params(8, 8, 1, direct_sample=50),
                api.make_params(8, 8, 1, direct_sample=2 | 48), api.make_params(8, 8, 0), api.make_params(0, 8, 1)):
        assert lib.kyhip_shard_float_count(C.byref(bad)) == A.KY_ERR_INVALID_VALUE


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="only meaningful on a box without a GPU")
def test_no_gpu_means_a_loud_error_not_a_fallback(A, api):
    scene = api.cornell_box_scene(A.CB_DEFAULT_SCENE, 16, 16)
    with pytest.raises(api.KyError) as e:
        api.render(scene, api.make_params(16, 16, 1))
    assert "no HIP device" in str(e.value) or "device" in str(e.value)
