import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ky_amd import api, _abi as A
scene = api.cornell_box_scene(A.CB_DEFAULT_SCENE, 4096, 4096)
p = api.make_params(4096, 4096, 256, max_path_depth=16)
t = time.time(); img = api.render(scene, p); dt = time.time() - t
print("C5 geometry 4096x4096x256spp d16: kernel %.1f ms (%.2f Gsamples/s), call %.2f s, mean %.5f, finite %s" % (api.kernel_ms(), 4096 * 4096 * 256 / api.kernel_ms() / 1e6, dt, img.mean(), np.isfinite(img).all()))
small = api.render(api.cornell_box_scene(A.CB_DEFAULT_SCENE, 256, 256), api.make_params(256, 256, 256, max_path_depth=16))
print("256x256 mean %.5f" % small.mean())
