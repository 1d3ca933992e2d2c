"""queue-engine scheduling statistics (library built with -DKY_QE_STATS): batches and lanes per state, polls."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ky_amd import api, _abi as A
lib = A.load_kyhip()
lib.kyhip_set_engine(1)
w, h, spp = 1024, 768, int(sys.argv[1]) if len(sys.argv) > 1 else 64
scene = api.mis_scene(w, h) if "veach" in sys.argv else api.cornell_box_scene(A.CB_DEFAULT_SCENE, w, h)
p = api.make_params(w, h, spp)
out = (C.c_ulonglong * 32)()
api.render(scene, p)
lib.kyhip_debug_stats(out, 1)
api.render(scene, p)
lib.kyhip_debug_stats(out, 1)
names = ["REGEN", "TRACE", "NEE", "SHADOW", "CONT"]
ns = w * h * spp
tot = sum(out[16:21]) + out[24] + out[25]
for i, n in enumerate(names):
    b, l, c = out[i], out[8 + i], out[16 + i]
    print("%-7s batches %10d  lanes/batch %.1f  lanes/sample %.3f  clocks/batch %8.0f  share %.3f" % (n, b, l / max(b, 1), l / ns, c / max(b, 1), c / tot))
nb = sum(out[:5])
print("acquire clocks/batch %.0f share %.3f   push clocks/batch %.0f share %.3f   kernel %.2f ms" % (out[24] / nb, out[24] / tot, out[25] / nb, out[25] / tot, api.kernel_ms()))
