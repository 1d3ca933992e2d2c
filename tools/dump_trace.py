import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from ky_amd import api, _abi as A
from oracle import kyoracle as O
np.set_printoptions(linewidth=250, precision=7, suppress=False)
scene = api.cornell_box_scene(A.CB_DEFAULT_SCENE, 64, 64)
params = api.make_params(64, 64, 512, direct_sample=48)
x, y, s = 18, 50, 254
g = api.kat_li(scene, params, x, y, 0, 512); c = O.li(scene, params, x, y, 0, 512)
print("g", g[s], "c", c[s], "rel", np.abs(g[s]-c[s]).max()/np.abs(c[s]).max())
g_rows, g_li = api.kat_li_trace(scene, params, x, y, s); c_rows = O.trace_li(scene, params, x, y, s)
for k in range(len(g_rows)):
    print("G", k, g_rows[k]); print("C", k, c_rows[k]); print("d", k, g_rows[k] - c_rows[k])
