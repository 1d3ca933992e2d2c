#!/usr/bin/env python3
"""Kernel time by strategy / scene (ablation of the NEE halves)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ky_amd import api, _abi as A
W, H, spp = 1024, 768, 128
scene = api.cornell_box_scene(A.CB_DEFAULT_SCENE, W, H)
for strat, name in ((0, "idle"), (16, "bsdf_mis"), (32, "light_mis"), (48, "both_mis"), (4, "bsdf"), (8, "light")):
    p = api.make_params(W, H, spp, direct_sample=strat)
    api.render(scene, p); api.render(scene, p)
    ms = api.kernel_ms()
    print("cornell %-10s kernel %7.2f ms  %8.1f Msamples/s" % (name, ms, W * H * spp / ms / 1e3))
for d in (1, 2, 3, 5, 8, 16):
    p = api.make_params(W, H, spp, max_path_depth=d)
    api.render(scene, p); api.render(scene, p)
    ms = api.kernel_ms()
    print("cornell both_mis depth %2d kernel %7.2f ms  %8.1f Msamples/s" % (d, ms, W * H * spp / ms / 1e3))
for integ, name in ((0, "position"), (6, "direct")):
    p = api.make_params(W, H, spp, integrator=integ)
    api.render(scene, p); api.render(scene, p)
    ms = api.kernel_ms()
    print("cornell %-10s kernel %7.2f ms  %8.1f Msamples/s" % (name, ms, W * H * spp / ms / 1e3))
