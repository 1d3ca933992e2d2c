"""Sanitizers on the CPU build.  The host code of libkyhip.so that needs no GPU -- ky_pack.cpp (scene packing, the occluder proof,
HostPool, the seam's lock order) and ky_jit.cpp (the run-time instantiations' code cache and its posix_spawn) -- is plain C++; `make sanitize-build` builds it
with g++ -fsanitize=address,undefined and -fsanitize=thread (plus ky_hostcheck.cpp: entry points for what has no C-ABI entry of its own).  These tests build
what they need on demand and run the checks in CHILD processes that have the sanitizer runtime preloaded; a report of any sanitizer fails them.
`make sanitize` additionally runs this whole CPU suite under the address build and writes profiles/<round>_sanitize_summary.txt.  Nothing here touches a GPU
(the pool has no GPU sanitizers)."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = os.path.join(ROOT, "build", "san")
FAKE_CC = os.path.join(ROOT, "tools", "sanitize", "fake_hipcc.sh")
REPORTS = ("ERROR: AddressSanitizer", "ERROR: LeakSanitizer", "runtime error:", "WARNING: ThreadSanitizer")

pytestmark = pytest.mark.skipif(os.environ.get("KY_SANITIZE") is not None, reason="already inside `make sanitize`'s sanitized pytest run")


def _make(*targets):
    r = subprocess.run(["make", "-s", "-j4", "-C", ROOT] + [os.path.join("build", "san", t) for t in targets], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


def _clean(out):
    for mark in REPORTS:
        assert mark not in out, out[-4000:]


def _gxx_lib(name):
    return subprocess.check_output(["g++", "-print-file-name=" + name], text=True).strip()


@pytest.mark.parametrize("san, options", [("asan", {"ASAN_OPTIONS": "detect_leaks=1"}), ("tsan", {"TSAN_OPTIONS": "halt_on_error=1 die_after_fork=0"})])
def test_stress_of_hostpool_seam_locks_and_code_cache(san, options, tmp_path):
    """Two caller threads x two "devices" taking the seam mutexes in lock_seams' order (what kyhip_render_multi does) with a HostPool job inside, a fork in between
    whose child runs the pool again; the banded add; every chunk schedule; six threads on the code cache (blocking and asynchronous requests, a stand-in compiler)."""
    _make("stress_" + san)
    env = dict(os.environ, KYHIP_CACHE_DIR=str(tmp_path / "cache"), KYHIP_HIPCC=FAKE_CC, **options)
    r = subprocess.run([os.path.join(SAN, "stress_" + san), "400"], capture_output=True, text=True, env=env, timeout=600)
    out = r.stdout + r.stderr
    _clean(out)
    assert r.returncode == 0 and "jit_stress -> 24 objects of 24 requests" in out and "seam_stress(400) -> 0" in out, out[-2000:]


CHILD = textwrap.dedent('''
    import ctypes as C, os, sys
    import numpy as np
    sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
    from ky_amd import _abi as A, api
    from oracle import kyoracle as O
    from test_random_scenes_gpu import random_room
    lib = A.load_kyhip()
    assert lib.kyhip_device_count() == 0 and A.SANITIZE == "asan"

    def pack(scene):
        feat, ph, ih, nocc, ts = C.c_int(), C.c_uint64(), C.c_uint64(), C.c_int(), C.c_int()
        rc = lib.kyhostcheck_pack(api._scene_ptr(scene), C.byref(feat), C.byref(ph), C.byref(ih), C.byref(nocc), C.byref(ts))
        assert rc == 0, (rc, lib.kyhip_last_error())
        return feat.value, ph.value, ih.value, nocc.value, ts.value

    # the shipped scenes: facts, occluder tables, the two-stage light -- and that packing is a pure function of the scene (same hashes twice)
    cornell = api.cornell_box_scene(A.CB_DEFAULT_SCENE, 1024, 768)
    f = pack(cornell)
    assert f == pack(cornell) and f[0] == 1 + 2 + 4 + 128 + 256 + 512 + 1024 + 2048 and f[3] == 5 and f[4] == 0, f      # one rectangle lamp that is its own carrier, walls and lamp housing are boxes, every planar surface in an axis plane; five walls proved away; two-stage scan
    veach = api.mis_scene(1280, 720)
    fv = pack(veach)
    assert fv[0] == 32 + 4 + 64 + 128 + 2048 + 4096 and fv[4] == -1, fv      # (4096: its tilted rectangles are planks about the x axis)
    for flag in (A.CB_LIGHT_POINT, A.CB_LIGHT_DIRECTION, A.CB_LIGHT_ENVIRONMENT):
        pack(api.cornell_box_scene(A.CB_BOTH_SMALL_SPHERES | flag, 256, 256))
    for seed in range(24):                       # rooms nobody tuned for: tilted walls, triangles, disks, every light kind (tests/test_random_scenes_gpu.py)
        room, kinds = random_room(A, api, O, seed, seed %% 2 == 1, 64, 48)
        pack(room)
        left = (C.c_int * 256)()
        assert lib.kyhip_scene_non_occluders(api._scene_ptr(room), -1, left, 256) >= 0
    # a scene at the ABI's limits, and broken scenes: errors, not overruns
    bad = A.Scene.from_buffer_copy(cornell.c)
    bad.surface_count = 10 ** 6
    assert lib.kyhostcheck_pack(C.byref(bad), None, None, None, None, None) == A.KY_ERR_LIMIT
    bad = A.Scene.from_buffer_copy(cornell.c)
    bad.environment_light = 7
    assert lib.kyhostcheck_pack(C.byref(bad), None, None, None, None, None) == A.KY_ERR_INVALID_VALUE
    # shard geometry and chunk schedules at the edges of their ranges
    for spp in list(range(1, 300)) + [447, 448, 449, 1024, 4096, 16384, (1 << 24)]:
        assert lib.kyhostcheck_chunks(spp) >= 1, spp
    p = api.make_params(16384, 16384, 1)
    assert lib.kyhostcheck_shard(C.byref(p)) > 0
    p = api.make_params(32767, 32767, 1)
    assert lib.kyhostcheck_shard(C.byref(p)) == A.KY_ERR_LIMIT      # 3.2e9 accumulator words: beyond the device's 32-bit pixel indices
    p = api.make_params(4096, 4096, 1 << 24)
    assert lib.kyhostcheck_shard(C.byref(p)) == A.KY_ERR_LIMIT
    p = api.make_params(4096, 4096, 16384); p.tile_first, p.tile_step = 7, 8
    assert lib.kyhostcheck_shard(C.byref(p)) > 0
    assert lib.kyhostcheck_add_rows(253, 61, 260, 4, 3) == 0 and lib.kyhostcheck_add_rows(8, 1, 8, 3, 2) == 0
    # the oracle (address + undefined build) on a small frame, and the host mirror's scene graph through its C API
    film = O.render(api.cornell_box_scene(A.CB_DEFAULT_SCENE, 48, 36), api.make_params(48, 36, 4))
    assert np.isfinite(film).all() and film.mean() > 0.01
    print("CHILD-OK")
''')


def test_packing_geometry_and_oracle_under_asan(tmp_path):
    """pack_scene / find_non_occluders / scene_input on the shipped scenes, 24 random rooms and broken scenes; shard geometry and chunk schedules at the edges of their
    ranges; the banded add; the oracle and ky.hpp's scene graph -- in a child Python with libasan + libubsan preloaded and the sanitizer builds loaded (KY_SANITIZE)."""
    _make("libkyhip_host_asan.so", "libkyhost_asan.so", "libkyoracle_asan.so")
    env = dict(os.environ, KY_SANITIZE="asan", LD_PRELOAD=_gxx_lib("libasan.so") + " " + _gxx_lib("libubsan.so"),
               ASAN_OPTIONS="detect_leaks=0", UBSAN_OPTIONS="print_stacktrace=1", KYHIP_CACHE_DIR=str(tmp_path / "cache"))
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], capture_output=True, text=True, env=env, timeout=900)
    out = r.stdout + r.stderr
    _clean(out)
    assert r.returncode == 0 and "CHILD-OK" in out, out[-3000:]


RACER = textwrap.dedent('''
    import ctypes as C, os, sys
    sys.path.insert(0, %(root)r)
    from ky_amd import _abi as A
    lib = A.load_kyhip()
    n = lib.kyhip_jit_compile(b"render_kernel<false, 48, false, false, 391, 11, false>")
    print("RACER", n, lib.kyhip_jit_status().decode())
    sys.exit(0 if n > 4000 else 1)
''')


def test_processes_share_a_cold_cache(tmp_path):
    """several ranks starting on a cold cache must not truncate each other's source files or compile the same object twice.  Four processes
    (address build) ask for the same instantiation at once through a stand-in compiler that logs every invocation: every process gets the object, the
    compiler ran ONCE (flock + re-check), the source files were written once, no temporary files are left."""
    _make("libkyhip_host_asan.so")
    cache = tmp_path / "cache"
    log = tmp_path / "cc.log"
    cc = tmp_path / "cc.sh"
    cc.write_text("#!/bin/bash\necho run >> %s\nexec %s \"$@\"\n" % (log, FAKE_CC))
    cc.chmod(0o755)
    env = dict(os.environ, KY_SANITIZE="asan", LD_PRELOAD=_gxx_lib("libasan.so") + " " + _gxx_lib("libubsan.so"), ASAN_OPTIONS="detect_leaks=0",
               KYHIP_CACHE_DIR=str(cache), KYHIP_HIPCC=str(cc))
    procs = [subprocess.Popen([sys.executable, "-c", RACER % {"root": ROOT}], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env) for _ in range(4)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for p, out in zip(procs, outs):
        _clean(out)
        assert p.returncode == 0, out[-2000:]
    assert log.read_text().count("run") == 1, (log.read_text(), outs)
    names = sorted(os.listdir(cache))
    assert sum(n.endswith(".hsaco") for n in names) == 1 and not any(".tmp" in n or n.endswith(".log") for n in names), names
    src = [d for d in names if d.startswith("src-")]
    assert len(src) == 1
    files = sorted(os.listdir(cache / src[0] / "ky_amd" / "csrc"))
    assert files == ["ky_device.hpp", "ky_render.hpp", "ky_scene.hpp", "ky_shard.hpp"], files     # the translation unit itself is removed after the compile


def test_compiler_never_sees_a_profilers_environment(tmp_path):
    """a profiler's preload must not travel into the compiler's processes, and a process that IS being profiled compiles nothing."""
    _make("libkyhip_host_asan.so")
    envlog = tmp_path / "env.log"
    cc = tmp_path / "cc.sh"
    cc.write_text("#!/bin/bash\nenv > %s\nexec %s \"$@\"\n" % (envlog, FAKE_CC))
    cc.chmod(0o755)
    preload = _gxx_lib("libasan.so") + " " + _gxx_lib("libubsan.so")
    base = dict(os.environ, KY_SANITIZE="asan", LD_PRELOAD=preload, ASAN_OPTIONS="detect_leaks=0", KYHIP_HIPCC=str(cc))
    r = subprocess.run([sys.executable, "-c", RACER % {"root": ROOT}], capture_output=True, text=True, env=dict(base, KYHIP_CACHE_DIR=str(tmp_path / "c1"), HSA_ENABLE_SDMA="0"))
    assert r.returncode == 0, r.stdout + r.stderr
    seen = envlog.read_text()
    assert "LD_PRELOAD" not in seen and "HSA_ENABLE_SDMA=0" in seen      # loader variables scrubbed, everything else passed on
    envlog.unlink()
    r = subprocess.run([sys.executable, "-c", RACER % {"root": ROOT}], capture_output=True, text=True,
                       env=dict(base, KYHIP_CACHE_DIR=str(tmp_path / "c2"), ROCPROFILER_REGISTER_ENABLED="1", HSA_TOOLS_LIB="librocprofiler-sdk-tool.so"))
    assert r.returncode == 1 and "stands down under a profiler" in r.stdout and not envlog.exists(), r.stdout + r.stderr
