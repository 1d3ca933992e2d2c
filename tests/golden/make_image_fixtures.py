#!/usr/bin/env python3
"""Generates tests/golden/reference_images.npz from the images the reference publishes (docs/images).

Run in the build container only (it reads /root/reference, which does not exist on the GPU box):
    python tests/golden/make_image_fixtures.py
The fixture holds DATA derived from the reference's published outputs -- 16x16 block means of the 8-bit,
gamma-2.2-encoded pixels scaled to [0, 1] -- never source text.  The images are mosaics written by the
reference's drivers (README.md:15-29):
    render_debug.png     render_debug()          1x3 cells of 512x308: Veach position / normal / basecolor, 10 spp
    veach_mis.jpg        render_mis_scene()      2x3 cells of 512x308: bsdf, light, idle / bsdf_mis, light_mis, both_mis, 10 spp
    multi_scene_mis.jpg  render_multiple_scene() 3x4 cells of 256x256: {bsdf, light, both_mis} x Cornell {point, direction, area, environment}
    lighting_enum.jpg    render_lighting_enum()  1x4 cells of 256x256 (dead driver, recursive integrator; kept for completeness)
"""
import os

import numpy as np
from PIL import Image

SRC = "/root/reference/docs/images"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_images.npz")


def block_means(name, block=16):
    im = np.asarray(Image.open(os.path.join(SRC, name)).convert("RGB"), np.float64) / 255.0
    h, w = (im.shape[0] // block) * block, (im.shape[1] // block) * block
    return im[:h, :w].reshape(h // block, block, w // block, block, 3).mean(axis=(1, 3)).astype(np.float32)


if __name__ == "__main__":
    data = {
        "render_debug": block_means("render_debug.png"),
        "veach_mis": block_means("veach_mis.jpg"),
        "multi_scene_mis": block_means("multi_scene_mis.jpg"),
        "lighting_enum": block_means("lighting_enum.jpg"),
    }
    np.savez_compressed(OUT, **data)
    for k, v in data.items():
        print(k, v.shape)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")
