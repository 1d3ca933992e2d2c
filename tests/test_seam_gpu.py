"""The host-film seam (kyhip_render / kyhip_render_multi: what a ky caller's integrator->render(&scene, sampler, &film) reaches) and the
host mirror's debug_pixel / debug_area (ky.cpp:3733-3787).  Round 4: the seam keeps its device buffers and a pinned staging film
between calls and adds the frame into the caller's film in row bands while the rest is still downloading; these tests pin that the
banded path is film_t::add_color (1586-1590) -- add, not overwrite, row strides honoured -- for every band count and frame size, and that
buffers sized by one call serve a larger or smaller next call."""
import numpy as np
import pytest

from test_random_scenes_gpu import random_room

pytestmark = pytest.mark.gpu


def test_host_film_is_added_to_in_bands(A, api, rng):
    """film += clamp01(frame), into a film with content and a row stride, at sizes that give 1, a few and the maximum number of bands."""
    for (w, h, spp) in [(24, 17, 4), (1024, 768, 2), (512, 300, 3), (1024, 768, 1), (40, 700, 2)]:   # grow, shrink, grow again: cached buffers
        scene = api.cornell_box_scene(A.CB_DEFAULT_SCENE, w, h)
        p = api.make_params(w, h, spp)
        frame = api.render(scene, p)                                     # into a zero film
        assert frame.min() >= 0 and frame.max() <= 1 and frame.max() > 0.2
        # the same frame into the middle of a larger film that already holds something
        big = rng.uniform(0, 1, (h + 9, w + 13, 3)).astype(np.float32)
        before = big.copy()
        api.render(scene, p, film=big, origin_px=(5, 3))
        assert np.array_equal(big[3:3 + h, 5:5 + w], before[3:3 + h, 5:5 + w] + frame), (w, h)
        rest, rest0 = big.copy(), before.copy()
        rest[3:3 + h, 5:5 + w] = 0
        rest0[3:3 + h, 5:5 + w] = 0
        assert np.array_equal(rest, rest0), "pixels outside the target were touched"
        # the device-list form of the same call
        multi = api.render_multi(scene, p, [0, 0, 0])
        assert np.array_equal(multi, frame)


def test_a_pinned_film_is_added_to_in_place(A, api, rng):
    """Round 5: a film in pinned host memory (kyhip_film_alloc: what ky.hpp's film_t allocates) is added to by the GPU's add kernel where it lies -- no staging
    film, no download, no host threads.  Same semantics as the banded path (add, row stride, nothing outside the target touched), bit-identical pixels, for the
    single-device call and the device-list call; a pageable film right afterwards takes the banded path again."""
    lib = A.load_kyhip()
    for (w, h, spp) in [(24, 17, 4), (1024, 768, 2), (40, 700, 2)]:
        scene = api.cornell_box_scene(A.CB_DEFAULT_SCENE, w, h)
        p = api.make_params(w, h, spp)
        frame = api.render(scene, p)                                     # pageable numpy film: the banded path
        assert b"host thread" in lib.kyhip_multi_status(0)
        pinned = api.PinnedFilm(h + 9, w + 13)
        pinned.array[...] = rng.uniform(0, 1, pinned.array.shape).astype(np.float32)
        before = pinned.array.copy()
        api.render(scene, p, film=pinned.array, origin_px=(5, 3))
        assert b"in place by the GPU" in lib.kyhip_multi_status(0), lib.kyhip_multi_status(0)
        assert np.array_equal(pinned.array[3:3 + h, 5:5 + w], before[3:3 + h, 5:5 + w] + frame), (w, h)
        rest, rest0 = pinned.array.copy(), before.copy()
        rest[3:3 + h, 5:5 + w] = 0
        rest0[3:3 + h, 5:5 + w] = 0
        assert np.array_equal(rest, rest0), "pixels outside the target were touched"
        whole = api.PinnedFilm(h, w)
        api.render_multi(scene, p, [0, 0, 0], film=whole.array)
        assert b"3 shard(s)" in lib.kyhip_multi_status(0) and b"in place" in lib.kyhip_multi_status(0)
        assert np.array_equal(whole.array, frame)
        del pinned, whole
    # the host mirror's film_t owns pinned pixels: integrator_t::render through the C++ API takes the in-place path by itself
    W, H = 48, 32
    film = api.render_host_api(api.cornell_box_scene(A.CB_DEFAULT_SCENE, W, H), 11, 5, 48, A.SAMPLER_RANDOM, 8, W, H)
    assert b"in place by the GPU" in lib.kyhip_multi_status(0) and film.max() > 0.2


def test_debug_pixel_and_debug_area_replay_the_render(A, api):
    """integrator_t::debug_pixel / debug_area (3733-3787) on the host mirror: a red frame is ADDED around the area and its pixels are
    cleared and rendered again; the replay (per-sample radiance, summed in sample order) is the pixel render() computed, to rounding."""
    W, H, spp = 48, 32, 24
    scene = api.cornell_box_scene(A.CB_DEFAULT_SCENE, W, H)
    film = api.render_host_api(scene, 11, 5, 48, A.SAMPLER_RANDOM, spp, W, H)
    rendered = film.copy()
    # debug_pixel
    out = api.debug_area_host_api(scene, 11, 5, 48, A.SAMPLER_RANDOM, spp, W, H, (20, 11), (21, 12), film=film.copy())
    assert np.allclose(out[11, 20], rendered[11, 20], atol=2e-6), (out[11, 20], rendered[11, 20])
    frame = np.zeros((H, W), bool)
    frame[10:13, 19:22] = True
    frame[11, 20] = False
    assert np.array_equal(out[frame], rendered[frame] + np.array([1, 0, 0], np.float32))       # add_color(color_t{1.f}): red is ADDED
    other = ~frame
    other[11, 20] = False
    assert np.array_equal(out[other], rendered[other])
    # debug_area at the film's corner: the part of the frame that would fall outside the film is skipped
    out = api.debug_area_host_api(scene, 11, 5, 48, A.SAMPLER_RANDOM, spp, W, H, (0, 0), (3, 2), film=rendered.copy())
    assert np.allclose(out[0:2, 0:3], rendered[0:2, 0:3], atol=2e-6)
    assert np.array_equal(out[2, 0:4], rendered[2, 0:4] + np.array([1, 0, 0], np.float32))
    assert np.array_equal(out[0:2, 3], rendered[0:2, 3] + np.array([1, 0, 0], np.float32))
    assert np.array_equal(out[3:], rendered[3:]) and np.array_equal(out[:, 4:], rendered[:, 4:])
    # another integrator and the debug sampler go the same way
    film = api.render_host_api(scene, 9, 5, 48, A.SAMPLER_DEBUG, 2, W, H)
    out = api.debug_area_host_api(scene, 9, 5, 48, A.SAMPLER_DEBUG, 2, W, H, (30, 20), (31, 21), film=film.copy())
    assert np.allclose(out[20, 30], film[20, 30], atol=2e-6)
    assert api.debug_area_host_api(scene, 7, 5, 48, A.SAMPLER_RANDOM, 1, W, H, (1, 1), (2, 2)) is None    # create_integrator -> nullptr (4638)


def test_shadow_ray_stacks_grow_with_the_launch(A, api, O, table_kernels):
    """Round-3 advice (high): the per-stream block of shadow-ray stacks was sized by the FIRST deferred-rays launch on the stream.  The
    fact-free variants run five workgroups per CU, the sphere-lights one six: a full grid of the second after the first wrote past the
    block.  Both at sizes that fill the chip, in that order, on one stream; the second image must equal its two-shard sum (bit for bit
    by construction) and sit on the oracle."""
    W, H, spp = 256, 256, 64
    sc = None
    for seed in range(64):   # a room with two or more lights that is not the sphere-lights case
        cand, kinds = random_room(A, api, O, seed, False, W, H)
        if len(kinds) >= 2:
            sc = cand
            break
    assert sc is not None
    p = api.make_params(W, H, spp)
    lib = A.load_kyhip()
    prev = lib.kyhip_set_shadow_queue(1)   # (left to itself the library traces this room's shadow rays inline: too few sphere lamps)
    try:
        room = api.render(sc, p)
        assert b"deferred shadow rays" in lib.kyhip_last_kernel(0) and (b"feat 0" in lib.kyhip_last_kernel(0) or not table_kernels), lib.kyhip_last_kernel(0)
        _stacks_grow(A, api, O, lib, sc, p, room, W, H, spp)
    finally:
        lib.kyhip_set_shadow_queue(prev)


def _stacks_grow(A, api, O, lib, sc, p, room, W, H, spp):
    veach = api.mis_scene(W, H)
    full = api.render(veach, p)
    assert b"deferred shadow rays" in lib.kyhip_last_kernel(0) and b"feat 6372" in lib.kyhip_last_kernel(0), lib.kyhip_last_kernel(0)
    halves = np.zeros_like(full)
    for r in range(2):
        api.render(veach, api.make_params(W, H, spp, tile_first=r, tile_step=2), film=halves)
    assert np.array_equal(full, halves)
    assert np.array_equal(room, api.render(sc, p))
    tile = O.render(veach, api.make_params(W, H, spp, tile_first=5, tile_step=64))
    mask = tile.max(axis=2) > 0
    assert mask.sum() > 500 and np.sqrt(np.mean((full[mask] - tile[mask]) ** 2)) < 5e-3
