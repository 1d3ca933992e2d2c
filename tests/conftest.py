import os
import sys

# OpenMP sizes its teams by the affinity mask; a box that shows 256 CPUs and grants two through its cgroup turns every oracle call into minutes of
# spinning against the quota (the two-rank bench test took 500 s there).  Set before numpy / torch bring an OpenMP runtime in.
def _cpus_granted():
    granted = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            granted = min(granted, max(1, int(round(int(quota) / int(period)))))
    except Exception:
        pass
    return granted


os.environ.setdefault("OMP_NUM_THREADS", str(_cpus_granted()))
# Run-time instantiations are ON by default in a single-process job (asynchronous: a scene outside the table switches kernels a few seconds in); the suite pins
# the table's kernels so that two renders of one scene are the same image, and tests/test_jit.py sets the modes it tests itself (KYHIP_JIT=1 runs the whole
# suite on instantiated kernels: the `table_kernels` fixture).
os.environ.setdefault("KYHIP_JIT", "0")

import numpy as np  # noqa: E402
import pytest  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """The in-tree libraries are built by __graft_entry__.build(); build them if a fresh checkout lacks them."""
    lib = os.path.join(ROOT, "ky_amd", "lib", "libkyhip.so")
    host = os.path.join(ROOT, "ky_amd", "lib", "libkyhost.so")
    orc = os.path.join(ROOT, "oracle", "libkyoracle.so")
    if not (os.path.exists(lib) and os.path.exists(host) and os.path.exists(orc)):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def A():
    from ky_amd import _abi
    return _abi


@pytest.fixture(scope="session")
def api():
    from ky_amd import api as _api
    return _api


@pytest.fixture(scope="session")
def O():
    from oracle import kyoracle
    kyoracle.load()
    return kyoracle


@pytest.fixture(scope="session")
def rng():
    return np.random.default_rng(20251001)


@pytest.fixture(scope="session")
def table_kernels():
    """False when the suite runs with KYHIP_JIT=1 (every launch on a kernel compiled for it): assertions about WHICH row of the library's table a launch
    picked are skipped then, every numeric assertion stays."""
    import os
    return os.environ.get("KYHIP_JIT", "0") in ("", "0")


@pytest.fixture
def no_boxes(A):
    """For tests that compare two KERNELS of the library bit for bit (or to a few ulp): the box traversal (kyhip_set_boxes) is the one switch that
    changes arithmetic -- a slab test's hit distance differs from the rectangle test's by up to 15 units in the last place -- and only some kernels have it, so
    these tests run without it; tests/test_boxes.py bounds what the switch itself moves."""
    lib = A.load_kyhip()
    prev = lib.kyhip_set_boxes(0)
    yield
    lib.kyhip_set_boxes(prev)
