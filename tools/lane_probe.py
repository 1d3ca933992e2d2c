#!/usr/bin/env python3
"""Lane-utilisation probe (debug build with -DKY_PROFILE_LANES): active lanes per visit at the main sites of path_step."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ky_amd import api, _abi as A
names = {0: "path trace", 1: "NEE entered", 2: "MIS bsdf trace", 3: "light sample", 4: "shadow trace", 5: "unoccluded eval", 6: "post-hit (vertex)", 7: "continuation sample", 8: "by_bsdf query (surface-parallel)", 9: "by_bsdf fast path calls", 10: "by_bsdf pending lanes"}
which = sys.argv[1] if len(sys.argv) > 1 else "cornell"
scene = api.cornell_box_scene(A.CB_DEFAULT_SCENE, 512, 384) if which == "cornell" else api.mis_scene(640, 360)
W, H = (512, 384) if which == "cornell" else (640, 360)
lib = A.load_kyhip()
buf = (C.c_ulonglong * 32)()
lib.kyhip_debug_lane_probe(buf)
api.render(scene, api.make_params(W, H, 64))
lib.kyhip_debug_lane_probe(buf)
n = W * H * 64
print(which, "samples", n)
for k, name in names.items():
    lanes, visits = buf[k], buf[k + 16]
    if visits:
        print("%-22s visits/sample*64 %8.3f  lane-visits/sample %7.3f  active lanes/visit %5.1f / 64" % (name, visits * 64 / n, lanes / n, lanes / visits))
