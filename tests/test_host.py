"""CPU: the host C++ layer restates the reference's workload data and film/camera semantics."""
import os
import struct

import numpy as np
import pytest


def v(a):
    return np.array(list(a), np.float64)


def test_cornell_scene_data(A, api):
    hs = api.cornell_box_scene(A.CB_DEFAULT_SCENE, 256, 256)
    s = hs.c
    assert (s.shape_count, s.material_count, s.light_count, s.surface_count, s.environment_light) == (13, 8, 1, 12, -1)
    # surface order decides ties (ky.cpp:3400-3426): left,right,top,bottom,back,left_ball,right_ball,left2,right2,front2,back2,bottom2
    shapes = [s.surfaces[i].shape for i in range(12)]
    assert shapes == [0, 1, 4, 3, 2, 6, 7, 8, 9, 10, 11, 12]
    mats = [s.surfaces[i].material for i in range(12)]
    assert mats == [3, 2, 1, 5, 4, 6, 7, 1, 1, 1, 1, 0]  # green, red, white, glossy, blue, mirror, glass, white x4, black
    assert [s.surfaces[i].area_light for i in range(12)] == [-1] * 11 + [0]
    assert s.lights[0].kind == A.LIGHT_AREA and s.lights[0].shape == 12 and list(s.lights[0].color) == [25, 25, 25]
    # values printed from the reference's own types (marked with a dagger)
    np.testing.assert_allclose(v(s.shapes[6].p[0]), [-0.538850009, -0.0245300531, -0.780019999], rtol=0, atol=1e-9)
    np.testing.assert_allclose(v(s.shapes[7].p[0]), [0.558309972, -0.0245300531, -0.780019999], rtol=0, atol=1e-9)
    np.testing.assert_allclose(v(s.shapes[5].p[0]), [0.00972998142, -0.0245300531, -0.480019987], rtol=0, atol=1e-9)
    assert s.shapes[6].radius == 0.5 and s.shapes[5].radius == np.float32(0.8)
    m = s.materials[5]
    assert m.kind == A.MATERIAL_PLASTIC and m.exponent == 90
    assert (m.diffuse_probability, m.specular_probability) == (0.125, 0.875)
    np.testing.assert_allclose(v(s.shapes[12].normal), [0, 0, -1], atol=1e-7)  # bottom2 emits downwards
    assert s.materials[7].eta == np.float32(1.6)


def test_cornell_light_variants_and_bounding_sphere(A, api, O):
    for flag, kind in ((A.CB_LIGHT_AREA, A.LIGHT_AREA), (A.CB_LIGHT_DIRECTION, A.LIGHT_DIRECTION), (A.CB_LIGHT_POINT, A.LIGHT_POINT),
                       (A.CB_LIGHT_ENVIRONMENT, A.LIGHT_ENVIRONMENT)):
        h = api.cornell_box_scene(A.CB_BOTH_SMALL_SPHERES | flag, 256, 256)
        s = h.c
        assert s.light_count == 1 and s.lights[0].kind == kind
        assert s.surface_count == (12 if flag == A.CB_LIGHT_AREA else 7)
        assert s.environment_light == (0 if flag == A.CB_LIGHT_ENVIRONMENT else -1)
        bs = O.world_bounding_sphere(h)  # the oracle's restatement of bounds3_t::bounding_sphere
        np.testing.assert_allclose(bs, [0.00972998142, -0.0245300531, 0, 2.21705961], rtol=0, atol=2e-7)
        if kind in (A.LIGHT_DIRECTION, A.LIGHT_ENVIRONMENT):  # host preprocess == oracle == reference value
            assert abs(s.lights[0].world_radius - 2.21705961) < 2e-7
    hs = api.cornell_box_scene(A.CB_BOTH_SMALL_SPHERES | A.CB_LIGHT_POINT, 64, 64)
    s = hs.c
    np.testing.assert_allclose(v(s.lights[0].color), [70 / (4 * np.pi)] * 3, rtol=1e-6)
    np.testing.assert_allclose(v(s.lights[0].position), [0, 0.5, 1])
    hs = api.cornell_box_scene(A.CB_BOTH_SMALL_SPHERES | A.CB_LIGHT_DIRECTION, 64, 64)
    s = hs.c
    np.testing.assert_allclose(v(s.lights[0].direction), np.array([-1, -1.5, -1]) / np.linalg.norm([-1, -1.5, -1]), rtol=1e-6)
    assert list(s.lights[0].color) == [10, 4, 0]


def test_both_large_spheres_is_an_error(A, api):
    with pytest.raises(api.KyError) as e:  # ky.cpp:3268-3271
        api.cornell_box_scene(A.CB_LARGE_MIRROR | A.CB_LARGE_GLASS | A.CB_LIGHT_AREA, 64, 64)
    assert "both large balls" in str(e.value)
    hs = api.cornell_box_scene(A.CB_LARGE_GLASS | A.CB_LIGHT_AREA, 64, 64)
    s = hs.c
    assert s.surface_count == 11 and s.surfaces[5].shape == 5 and s.surfaces[5].material == 7


def test_veach_scene_data_and_cross_bound_lights(A, api):
    hs = api.mis_scene(1280, 720)
    s = hs.c
    assert (s.shape_count, s.material_count, s.light_count, s.surface_count) == (11, 3, 5, 11)
    # lights 1 and 2 SAMPLE each other's spheres (ky.cpp:3498-3499) but the surfaces carry them straight (3525-3526)
    assert [s.lights[i].shape for i in range(5)] == [6, 8, 7, 9, 10]
    assert [s.surfaces[i].area_light for i in range(11)] == [-1] * 6 + [0, 1, 2, 3, 4]
    np.testing.assert_allclose([s.lights[i].color[0] for i in range(5)], [800, 901.803, 100, 11.1111, 1.23457], rtol=1e-6)
    np.testing.assert_allclose([s.shapes[i].radius for i in range(6, 11)], [0.5, 0.03333, 0.1, 0.3, 0.9], rtol=1e-6)
    m = s.materials[2]
    np.testing.assert_allclose([m.diffuse_probability, m.specular_probability], [0.0814170763, 0.918582976], rtol=0, atol=1e-8)
    assert m.exponent == 5000
    for i in range(6):  # every rectangle was built with flip_normal = true
        assert s.shapes[i].kind == A.SHAPE_RECTANGLE
    np.testing.assert_allclose(v(s.shapes[0].normal), [0, 1, 0], atol=1e-7)   # floor faces up after the flip
    np.testing.assert_allclose(v(s.shapes[1].normal), [0, 0, 1], atol=1e-7)   # back wall: stored normal +z; hits report the ray-facing side (1289)


def test_camera_known_rays(A, api, O):
    hs = api.cornell_box_scene(A.CB_DEFAULT_SCENE, 256, 256)
    s = hs.c
    r = O.kat_camera(s.camera, np.array([[128, 128], [0, 0]], np.float32))
    np.testing.assert_allclose(r[0, 3:], [0.00688625313, -0.998505473, -0.0542161278], rtol=0, atol=1e-8)
    np.testing.assert_allclose(r[1, 3:], [-0.354752183, -0.880776882, 0.313660711], rtol=0, atol=1e-7)
    np.testing.assert_allclose(r[0, :3], [-0.0439815, 4.12529, 0.222539], rtol=1e-7)
    hs = api.mis_scene(1280, 720)
    s = hs.c
    r = O.kat_camera(s.camera, np.array([[640, 360]], np.float32))
    np.testing.assert_allclose(r[0, 3:], [0, -0.304775715, 0.952424109], rtol=0, atol=2e-7)  # 2 ulp: the survey build (GCC) normalises via double sqrt


def test_film_writers(A, api, tmp_path):
    host = A.load_kyhost()
    assert [host.kyhost_gamma_encoding(x) for x in (-1.0, 0.0, 0.5, 1.0, 7.0)] == [0, 0, int(0.5 ** (1 / 2.2) * 255 + .5), 255, 255]
    rgb = np.zeros((2, 4, 3), np.float32)
    rgb[0, 0] = [1, 0, 0]
    rgb[1, 3] = [0, 0.5, 1]
    f = str(tmp_path / "t.bmp")
    api.store_image(f, rgb, "bmp")
    b = open(f, "rb").read()
    assert b[:2] == b"BM" and len(b) == 54 + 2 * 4 * 3
    size, _, off, ih, w, h, planes, bpp = struct.unpack("<IIIIiihh", b[2:30])
    assert (size, off, ih, w, h, planes, bpp) == (54 + 24, 54, 40, 4, 2, 1, 24)
    body = np.frombuffer(b[54:], np.uint8).reshape(2, 4, 3)  # bottom-up, BGR (ky.cpp:1722-1733)
    assert list(body[1, 0]) == [0, 0, 255] and list(body[0, 3]) == [255, 186, 0]
    f = str(tmp_path / "t.ppm")
    api.store_image(f, rgb, "ppm")
    assert open(f).read().startswith("P3\n4 2\n255\n255 0 0 ")
    f = str(tmp_path / "t.hdr")
    api.store_image(f, rgb, "hdr")
    raw = open(f, "rb").read()
    head = b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 2 +X 4\n"
    assert raw.startswith(head) and len(raw) == len(head) + 8 * 4
    assert list(raw[len(head):len(head) + 4]) == [128, 0, 0, 129]  # (1,0,0): mantissa 0.5*256, exponent 1+128


def test_written_files_decode_back_to_the_film(A, api, tmp_path):
    """The writers checked from the READER's side, with no restatement of ky.cpp in between.  HDR: the Radiance format's published decoding
    (value = byte * 2^(exponent - 128 - 8), the formula ky.cpp:1762-1765 quotes) applied to the product's file must bracket the film:
    the writer truncates (1772-1774), so byte * f <= value < (byte + 1) * f.  PPM: every token of the P3 body is the gamma code whose
    reference-produced values tests/golden/rewrite_kat.npz holds (gamma_in / gamma_out: GammaEncoding of smallpt_rewrite.cpp:494 == ky.cpp:1548)."""
    rng = np.random.default_rng(11)
    film = (rng.uniform(0, 1, (7, 9, 3)) ** 6 * 300).astype(np.float32)
    film[0, 0] = 0
    film[1, 1] = [1e-35, 0, 0]      # below the writer's 1e-32 threshold: stored as four zero bytes
    film[2, 2] = [5.0, 1e-4, 0.0]   # one component dominates: the others lose their bits, not their bracket
    f = str(tmp_path / "d.hdr")
    api.store_image(f, film, "hdr")
    raw = open(f, "rb").read()
    head, _, body = raw.partition(b"\n\n")
    assert head.split(b"\n")[0] == b"#?RADIANCE"
    res, _, pix = body.partition(b"\n")
    tok = res.split()
    assert tok[0] == b"-Y" and tok[2] == b"+X"
    h, w = int(tok[1]), int(tok[3])
    assert (h, w) == film.shape[:2]
    q = np.frombuffer(pix, np.uint8).reshape(h, w, 4).astype(np.float64)
    scale = np.ldexp(1.0, q[..., 3].astype(np.int64) - 136)[..., None]
    lo, hi = q[..., :3] * scale, (q[..., :3] + 1) * scale
    stored = q[..., 3] > 0
    v = film.astype(np.float64)
    assert np.all(lo[stored] <= v[stored] * (1 + 1e-6)) and np.all(v[stored] < hi[stored] * (1 + 1e-6))
    assert np.all(q[~stored] == 0) and np.all(v[~stored].max(axis=-1) < 1e-32)
    assert stored.sum() == h * w - 2

    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rewrite_kat.npz"))
    gin, gout = G["gamma_in"], G["gamma_out"]
    n = (len(gin) // 3) * 3
    strip = gin[:n].reshape(1, n // 3, 3)
    f = str(tmp_path / "d.ppm")
    api.store_image(f, strip, "ppm")
    words = open(f).read().split()
    assert words[:4] == ["P3", str(n // 3), "1", "255"]
    assert np.array_equal(np.array(words[4:], np.int64), gout[:n].astype(np.int64))
