"""Lane-engine phase clocks (library built with -DKY_PROFILE_CLOCKS): share of the wave time per phase of the loop."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ky_amd import api, _abi as A
lib = A.load_kyhip()
w, h, spp = 1024, 768, int(sys.argv[1]) if len(sys.argv) > 1 else 64
scene = api.mis_scene(w, h) if "veach" in sys.argv else api.cornell_box_scene(A.CB_DEFAULT_SCENE, w, h)
p = api.make_params(w, h, spp)
out = (C.c_ulonglong * 16)()
api.render(scene, p); lib.kyhip_debug_clocks(out, 1)
api.render(scene, p); lib.kyhip_debug_clocks(out, 1)
names = {0: "item bookkeeping / flush", 1: "regenerate + trace (+retrace)", 2: "vertex setup (bsdf, frame)", 3: "light loop: 4 random numbers",
         4: "estimate_by_bsdf (MIS bsdf ray)", 5: "light sample", 6: "shadow traversal", 7: "bsdf eval + weights", 8: "light loop tail / Lo update",
         9: "continuation: bsdf sample, roulette"}
tot = sum(out)
for k, n in names.items():
    print("%-36s %6.3f" % (n, out[k] / tot))
print("kernel %.2f ms (with probes)" % api.kernel_ms())
