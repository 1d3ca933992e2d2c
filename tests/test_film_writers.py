"""the film output path.  The product's C++ writers (ky_amd/host/ky.hpp: gamma_encoding, store_ppm_impl,
store_bmp_impl, store_hdr_impl) against an independent numpy restatement of ky.cpp:1548, 1646-1782 (oracle/film_writers.py),
byte for byte; the numpy BMP routine itself is pinned by the file the REFERENCE BINARY writes (smallpt_rewrite.cpp carries
the same store_bmp_impl)."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

from oracle import film_writers as FW

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_EXE = os.path.join(ROOT, "oracle", "_ref", "smallpt_rewrite")
GOLD = os.path.join(ROOT, "tests", "golden", "smallpt_rewrite_16.npz")


def _films(rng):
    out = []
    for (h, w) in ((2, 4), (3, 5), (7, 2), (1, 1), (9, 13), (16, 64)):    # widths that are and are not multiples of 4
        f = rng.uniform(-0.2, 1.3, (h, w, 3)).astype(np.float32)
        f[0, 0] = [0, 1, 0.5]
        f.reshape(-1)[::7] *= np.float32(1e-3)                            # small values: the steep end of the gamma curve
        out.append(f)
    big = (rng.uniform(0, 1, (5, 6, 3)) ** 8 * 1e4).astype(np.float32)     # HDR range for the RGBE writer
    big[0, 0] = 0
    big[1, 1] = [1e-33, 0, 0]
    out.append(big)
    return out


def test_gamma_encoding_every_level(A):
    host = A.load_kyhost()
    xs = np.concatenate([np.linspace(-0.5, 1.5, 4001), (np.arange(256) / 255.0) ** 2.2, [0.0, 1.0, 1e-9, 0.999999]]).astype(np.float32)
    want = FW.gamma_encoding(xs)
    got = np.array([host.kyhost_gamma_encoding(float(x)) for x in xs], np.uint8)
    assert np.array_equal(got, want)
    assert want.min() == 0 and want.max() == 255 and FW.gamma_encoding(np.float32(0.5)) == int(0.5 ** (1 / 2.2) * 255 + .5)


@pytest.mark.parametrize("kind", ["bmp", "ppm", "hdr"])
def test_cpp_writers_match_numpy_restatement(api, rng, tmp_path, kind):
    fn = {"bmp": FW.bmp_bytes, "ppm": FW.ppm_bytes, "hdr": FW.hdr_bytes}[kind]
    for i, film in enumerate(_films(rng)):
        path = str(tmp_path / ("f%d.%s" % (i, kind)))
        api.store_image(path, film, kind)
        got = open(path, "rb").read()
        want = fn(film)
        assert got == want, (kind, film.shape, len(got), len(want))


def test_bmp_padding_quirk():
    """Header advertises padded rows, body is unpadded (ky.cpp:1669-1670 vs 1728-1733)."""
    b = FW.bmp_bytes(np.zeros((3, 5, 3), np.float32))
    assert int.from_bytes(b[2:6], "little") == 54 + 16 * 3 and len(b) == 54 + 15 * 3


def test_numpy_bmp_header_is_the_reference_binarys():
    g = np.load(GOLD)
    mine = FW.bmp_bytes(np.zeros((768, 1024, 3), np.float32))
    assert mine[:54] == g["header"].tobytes() and len(mine) == int(g["file_bytes"])


@pytest.mark.skipif(not os.path.exists(REF_EXE), reason="oracle/_ref/smallpt_rewrite not built (needs /root/reference)")
def test_numpy_bmp_reproduces_the_reference_binarys_file(A, O):
    """Whole file, every byte: the reference renders and writes image.bmp; the oracle renders the same film (byte-exact
    restatement, tests/test_smallpt_rewrite.py) and the numpy writer encodes it."""
    with tempfile.TemporaryDirectory() as d:
        subprocess.check_call([REF_EXE, "16"], cwd=d, stderr=subprocess.DEVNULL)
        ref = open(os.path.join(d, "image.bmp"), "rb").read()
    film = O.sprw_render(O.sprw_scene(), A.SmallptParams(1024, 768, 4, 1234, 10, A.SP_VARIANT_REWRITE), rng_mode=1)
    mine = FW.bmp_bytes(film)          # a float64 film: smallpt_rewrite's Float is double, GammaEncoding identical otherwise
    assert len(mine) == len(ref)
    diff = np.flatnonzero(np.frombuffer(mine, np.uint8) != np.frombuffer(ref, np.uint8))
    assert diff.size == 0, diff[:10]
