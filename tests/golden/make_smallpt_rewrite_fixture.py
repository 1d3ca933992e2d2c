#!/usr/bin/env python3
"""Generates tests/golden/smallpt_rewrite_16.npz from the REFERENCE ITSELF, run in the container that holds /root/reference:

    make -C oracle ref                       # g++ -std=c++20 -O2 -fopenmp /root/reference/smallpt2pbrt/smallpt_rewrite.cpp (unmodified)
    python tests/golden/make_smallpt_rewrite_fixture.py

The reference renders 1024 x 768 at argv[1] / 4 samples per pixel into image.bmp (gamma 2.2, 8 bit, BGR, bottom-up;
smallpt_rewrite.cpp:494, 534-606, 1383-1404).  Kept (small, data only):
  sha256      of the pixel bytes of the argv = 16 (4 spp) image, top-down RGB
  band        rows 352..383 of that image (32 x 1024 x 3 bytes: through the mirror and glass spheres)
  means64     16 x 16 block means of the 8-bit image of a 256-spp run (argv = 1024; about 2 minutes on 8 cores), for the
              statistical comparison of renders that use other random numbers
  header      the 54 header bytes of the BMP file (pins the writer: film_t::store_bmp_impl in ky.cpp is the same routine)
"""
import hashlib
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
EXE = os.path.join(ROOT, "oracle", "_ref", "smallpt_rewrite")


def run_reference(argv1):
    with tempfile.TemporaryDirectory() as d:
        subprocess.check_call([EXE, str(argv1)], cwd=d, stderr=subprocess.DEVNULL)
        raw = open(os.path.join(d, "image.bmp"), "rb").read()
    return raw


def decode_bmp(raw):
    off = int.from_bytes(raw[10:14], "little")
    w = int.from_bytes(raw[18:22], "little", signed=True)
    h = int.from_bytes(raw[22:26], "little", signed=True)
    px = np.frombuffer(raw[off:off + w * h * 3], np.uint8).reshape(h, w, 3)
    return np.ascontiguousarray(px[::-1, :, ::-1])   # bottom-up BGR -> top-down RGB


def main():
    if not os.path.exists(EXE):
        sys.exit("build the reference first: make -C oracle ref")
    raw16 = run_reference(16)
    img16 = decode_bmp(raw16)
    raw_hi = run_reference(1024)
    img_hi = decode_bmp(raw_hi).astype(np.float64)
    means = img_hi.reshape(48, 16, 64, 16, 3).mean(axis=(1, 3))
    out = os.path.join(ROOT, "tests", "golden", "smallpt_rewrite_16.npz")
    np.savez_compressed(out, sha256=np.frombuffer(hashlib.sha256(img16.tobytes()).digest(), np.uint8), band=img16[352:384],
                        means64=means.astype(np.float32), header=np.frombuffer(raw16[:54], np.uint8), file_bytes=np.int64(len(raw16)))
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
