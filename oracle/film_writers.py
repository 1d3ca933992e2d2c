"""Independent numpy restatement of the reference's film output path: gamma_encoding (ky.cpp:1548) and the
three writers store_ppm_impl (1646-1659), store_bmp_impl (1661-1737), store_hdr_impl (1739-1782).

TEST INFRASTRUCTURE ONLY (like everything under oracle/): tests compare the bytes these functions produce with the files the
product's C++ writers (ky_amd/host/ky.hpp) write, and -- for the BMP routine, which smallpt2pbrt/smallpt_rewrite.cpp:534-606
shares with ky.cpp -- with the file the reference binary itself writes (oracle/_ref/smallpt_rewrite).  Nothing here is
imported by ky_amd/.
"""
import struct

import numpy as np


def gamma_encoding(x):
    """uint8_t gamma_encoding(float_t x) { return pow(clamp01(x), 1 / 2.2) * 255 + .5; }  (1548): the clamp is std::clamp on the
    float, pow runs in double (the exponent 1 / 2.2 is a double), the conversion to uint8_t truncates."""
    x = np.asarray(x)
    c = np.clip(x, 0, 1).astype(np.float64)          # float -> double is exact
    return (np.power(c, 1 / 2.2) * 255 + .5).astype(np.int64).astype(np.uint8)


def ppm_bytes(rgb):
    """store_ppm_impl: "P3\\n{w} {h}\\n255\\n" then every float as a decimal number followed by one space (1650-1656)."""
    h, w, _ = rgb.shape
    body = " ".join(str(int(v)) for v in gamma_encoding(rgb).reshape(-1))
    return ("P3\n%d %d\n%d\n" % (w, h, 255) + body + " ").encode()


def bmp_bytes(rgb):
    """store_bmp_impl: "BM" + the 52-byte packed header struct (1677-1703) + the body, bottom row first, BGR (1718-1733).
    The header's file size counts rows padded to 4 bytes (1669-1670, 1699) but the rows are written UNPADDED (1728, 1732):
    for widths that are not a multiple of 4 the file is shorter than its header says -- the reference's quirk, kept."""
    h, w, ch = rgb.shape
    assert ch == 3
    padded_line = (w * ch + 3) & ~3
    header = b"BM" + struct.pack("<IIIIiihhIIIIII", 14 + 40 + padded_line * h, 0, 14 + 40, 40, w, h, 1, ch * 8, 0, 0, 0, 0, 0, 0)
    body = gamma_encoding(rgb)[::-1, :, ::-1]        # bottom-up, BGR
    return header + np.ascontiguousarray(body).tobytes()


def hdr_bytes(rgb):
    """store_hdr_impl: Radiance header (1743-1746), then 4 bytes per pixel: v = max(r, g, b); if v >= 1e-32f:
    m = float(frexp(v) * 256.f / v), bytes uint8(r * m), uint8(g * m), uint8(b * m), uint8(e + 128); else zeros (1752-1776)."""
    h, w, _ = rgb.shape
    px = np.asarray(rgb, np.float32).reshape(-1, 3)
    v = px.max(axis=1)
    mant, e = np.frexp(v.astype(np.float32))
    ok = v >= np.float32(1e-32)
    with np.errstate(divide="ignore", invalid="ignore"):
        # frexp(v, &e) returns a float (v is a float); * 256.f / v in float, then float_t(...)
        m = (mant.astype(np.float32) * np.float32(256.0) / v).astype(np.float32)
        q = (px * m[:, None]).astype(np.float32)
    out = np.zeros((px.shape[0], 4), np.uint8)
    out[ok, :3] = q[ok].astype(np.int64).astype(np.uint8)
    out[ok, 3] = (e[ok] + 128).astype(np.int64).astype(np.uint8)
    head = ("#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n" % (h, w)).encode()
    return head + out.tobytes()
