"""Run-time instantiations (kyhip_set_jit, include/kyhip.h): the library compiles the render kernel for a launch's exact template arguments from its
embedded source.  CPU part: the compile itself (a child process of the ROCm compiler, no GPU needed), the disk cache, the failure paths.  GPU part:
the instantiated kernels render what the table's kernels render."""
import ctypes as C
import os
import shutil
import time

import numpy as np
import pytest

HIPCC = os.environ.get("KYHIP_HIPCC") or ("/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else shutil.which("hipcc"))


@pytest.mark.skipif(not HIPCC, reason="no ROCm compiler on this machine")
def test_instantiations_compile_and_are_cached(A, tmp_path, monkeypatch):
    monkeypatch.setenv("KYHIP_CACHE_DIR", str(tmp_path / "cache"))
    lib = A.load_kyhip()
    # one rectangle area light with every fact of the Cornell box (1 + 2 + 4 + 128); deferred shadow rays with sphere lights; the debug sampler on general shapes
    sizes = {}
    for expr in (b"render_kernel<false, 48, false, false, 135, 11, false>", b"render_kernel<false, 32, true, false, 228, 11, false>", b"render_kernel<true, 16, false, true, 0, 10, true>"):
        t0 = time.time()
        n = lib.kyhip_jit_compile(expr)
        assert n > 4096, (expr, n, lib.kyhip_last_error())
        sizes[expr] = (n, time.time() - t0)
    assert b"on (" in lib.kyhip_jit_status()
    objects = [f for f in os.listdir(tmp_path / "cache") if f.endswith(".hsaco")]
    assert len(objects) == 3
    # a second request is served from memory; a fresh process would find the file (same name: hash of sources, flags and arguments)
    t0 = time.time()
    assert lib.kyhip_jit_compile(b"render_kernel<false, 48, false, false, 135, 11, false>") == sizes[b"render_kernel<false, 48, false, false, 135, 11, false>"][0]
    assert time.time() - t0 < 0.05
    head = open(tmp_path / "cache" / objects[0], "rb").read(24)
    assert head[:4] == b"\x7fELF" or head == b"__CLANG_OFFLOAD_BUNDLE__"
    # what is not a list of template arguments never reaches a command line; what does not compile is reported, not fatal
    assert lib.kyhip_jit_compile(b"render_kernel<0>; touch x>") == A.KY_ERR_INVALID_VALUE
    assert lib.kyhip_jit_compile(b"smallpt_kernel<1>") == A.KY_ERR_INVALID_VALUE
    assert lib.kyhip_jit_compile(b"render_kernel<false, 48, false, false, 7, 11, 3>") == A.KY_ERR_DEVICE
    assert b"failed" in lib.kyhip_jit_status()
    prev = lib.kyhip_set_jit(1)
    assert lib.kyhip_set_jit(prev) == 1


def test_a_missing_compiler_is_a_status_not_a_crash(A, tmp_path, monkeypatch):
    monkeypatch.setenv("KYHIP_CACHE_DIR", str(tmp_path / "cache2"))
    monkeypatch.setenv("KYHIP_HIPCC", str(tmp_path / "no-such-compiler"))
    lib = A.load_kyhip()
    assert lib.kyhip_jit_compile(b"render_kernel<false, 8, false, false, 0, 11, false>") == A.KY_ERR_DEVICE
    assert b"no-such-compiler" in lib.kyhip_jit_status()


@pytest.mark.gpu
def test_instantiated_kernels_render_what_the_table_renders(A, api, O, tmp_path, monkeypatch, no_boxes):
    from test_random_scenes_gpu import random_room
    from test_parity_gpu import general_shapes_scene
    monkeypatch.setenv("KYHIP_CACHE_DIR", str(tmp_path / "cache"))
    lib = A.load_kyhip()
    W, H = 64, 48
    room = None
    for seed in range(64):
        cand, kinds = random_room(A, api, O, seed, False, W, H)
        if len(kinds) >= 2:
            room = cand
            break
    cases = [(api.cornell_box_scene(A.CB_BOTH_SMALL_SPHERES | A.CB_LIGHT_POINT, W, H), api.make_params(W, H, 64), b", 3214, "),                   # the table has feat 8 for it (boxes off here); all of the scene's facts: 8 + 2 + 4 + 128 + 1024 + 2048
             (api.mis_scene(W, H), api.make_params(W, H, 32), None),                                                                # in the table with ALL of its facts (2276): nothing to compile
             (room, api.make_params(W, H, 64), b"48, false, false, "),                                                              # two lights, shadow rays inline, this room's facts (132 / 134; + 256 when its lamps are their own carriers)
             (room, api.make_params(W, H, 64), b"48, true, false, "),                                                               # the same with deferred rays (forced: kyhip_set_shadow_queue)
             (general_shapes_scene(A, api), api.make_params(48, 40, 64, direct_sample=A.DIRECT_LIGHT_MIS), b"32, "),               # the table: strategy at run time
             (api.cornell_box_scene(A.CB_DEFAULT_SCENE, W, H), api.make_params(W, H, 16, sampler=A.SAMPLER_DEBUG, integrator=A.INTEGRATOR_PATH_TRACING_RECURSION), b"true, 48")]
    prev = lib.kyhip_set_jit(0)
    try:
        for scene, p, mark in cases:
            lib.kyhip_set_shadow_queue(1 if (mark or b"").startswith(b"48, true") else -1)
            lib.kyhip_set_jit(0)
            table = api.render(scene, p)
            table_kernel = lib.kyhip_last_kernel(0)
            lib.kyhip_set_jit(1)
            own = api.render(scene, p)
            own_kernel = lib.kyhip_last_kernel(0)
            if mark is None:
                assert own_kernel == table_kernel and np.array_equal(own, table)
                continue
            assert b"run-time instantiation" in own_kernel and mark in own_kernel, (own_kernel, lib.kyhip_jit_status())
            # the same arithmetic in other surroundings: the compiler contracts a few multiply-adds differently (as between the table's own kernels)
            assert np.abs(own - table).max() < 2e-5 and table.max() > 0.1, (own_kernel, float(np.abs(own - table).max()))
            # deterministic, and sharding stays bit-identical under it
            parts = np.zeros_like(own)
            for r in range(3):
                q = A.RenderParams.from_buffer_copy(p)
                q.tile_first, q.tile_step = r, 3
                api.render(scene, q, film=parts)
            assert np.array_equal(parts, own)
        assert b"on (" in lib.kyhip_jit_status()
    finally:
        lib.kyhip_set_jit(prev)
        lib.kyhip_set_shadow_queue(-1)


@pytest.mark.gpu
def test_asynchronous_mode_renders_on_the_table_until_the_object_is_there(A, api, tmp_path, monkeypatch, no_boxes):
    """kyhip_set_jit(2) (round 5): a launch whose instantiation is not in the table does NOT wait for the compiler -- the table's kernel renders, a background
    thread compiles, and a later launch switches.  The first call returns in a fraction of the compile time, every frame is one of the two kernels' images
    (which differ by at most 2e-5), the switch happens within seconds, and a multi-shard call uses one kernel for all its shards."""
    monkeypatch.setenv("KYHIP_CACHE_DIR", str(tmp_path / "cache"))
    lib = A.load_kyhip()
    W, H = 96, 64
    scene = api.cornell_box_scene(A.CB_BOTH_SMALL_SPHERES | A.CB_LIGHT_AREA | A.CB_LIGHT_POINT, W, H)      # lamp + point light: no row of the table
    p = api.make_params(W, H, 32)
    prev = lib.kyhip_set_jit(0)
    try:
        table = api.render(scene, p)
        table_kernel = lib.kyhip_last_kernel(0)
        assert b"run-time" not in table_kernel
        lib.kyhip_set_jit(2)
        t0 = time.time()
        first = api.render(scene, p)
        dt_first = time.time() - t0
        name = lib.kyhip_last_kernel(0)
        assert name.split(b" [")[0] == table_kernel and b"is being compiled" in name and np.array_equal(first, table)     # nothing compiled yet: the table's kernel (which says so), the table's image
        assert dt_first < 1.0, dt_first                                                          # (a compile takes 2-3 s)
        own = None
        deadline = time.time() + 60
        while time.time() < deadline:
            img = api.render_multi(scene, p, [0, 0, 0])                                        # three shards per frame: all on one kernel
            if b"run-time instantiation" in lib.kyhip_last_kernel(0):
                own = img
                break
            assert np.array_equal(img, table)
            time.sleep(0.2)
        assert own is not None, lib.kyhip_jit_status()
        assert np.abs(own - table).max() < 2e-5 and lib.kyhip_jit_failures() == 0
        assert np.array_equal(api.render(scene, p), own)                                       # from here on: always the own kernel, deterministically
    finally:
        lib.kyhip_set_jit(prev)


def test_default_mode_policy(A):
    """Round 6: without KYHIP_JIT a single-process job with a compiler at hand and no profiler attached starts in mode 2 (asynchronous); KYHIP_JIT=0, a
    multi-process job (WORLD_SIZE > 1), a profiler's preload or a missing compiler start it in mode 0 -- and say why."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); from ky_amd import _abi as A; lib = A.load_kyhip(); "
            "print(lib.kyhip_set_jit(-1), '|', lib.kyhip_jit_status().decode())" % root)

    def run(**env):
        # (KY_SANITIZE: under `make sanitize` the children load the ordinary library, without the sanitizer runtime's preload this test strips with the others)
        e = {k: v for k, v in os.environ.items() if k not in ("KYHIP_JIT", "WORLD_SIZE", "LD_PRELOAD", "KYHIP_HIPCC", "KY_SANITIZE")}
        e.update(env)
        out = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        mode, why = out.stdout.strip().splitlines()[-1].split(" | ", 1)
        return int(mode), why

    mode, why = run()
    assert (mode, "on by default" in why) == ((2, True) if HIPCC and "/" in HIPCC else (0, False)), (mode, why)
    assert run(KYHIP_JIT="0")[0] == 0 and run(KYHIP_JIT="1")[0] == 1
    mode, why = run(WORLD_SIZE="8")
    assert mode == 0 and "multi-process" in why
    mode, why = run(ROCPROFSYS_MODE="trace")   # (a variable of a profiler that does nothing by itself: the policy looks at names)
    assert mode == 0 and "profiler" in why
    mode, why = run(KYHIP_HIPCC="no-such-compiler")
    assert mode == 0 and "compiler" in why
