import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
os.environ.setdefault("KYHIP_JIT", "0")
import numpy as np
from ky_amd import api, _abi as A
from oracle import kyoracle as O
np.set_printoptions(linewidth=250, precision=7, suppress=False)
scene = api.cornell_box_scene(A.CB_DEFAULT_SCENE, 1024, 768)
params = api.make_params(1024, 768, 1024)
x, y, s = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
g = api.kat_li(scene, params, x, y, 0, 1024); c = O.li(scene, params, x, y, 0, 1024)
print("g", g[s], "c", c[s])
g_rows, g_li = api.kat_li_trace(scene, params, x, y, s); c_rows = O.trace_li(scene, params, x, y, s)
print("rows", len(g_rows), len(c_rows), "g_li", g_li)
names = "bounce surf lobe px py pz nx ny nz wox woy woz bx by bz Lx Ly Lz fx fy fz pdf cos flags dB dL".split()
for k in range(min(len(g_rows), len(c_rows))):
    print("--- vertex", k)
    for j, n in enumerate(names):
        a, b = g_rows[k][j], c_rows[k][j]
        flag = "" if (a == b or abs(a - b) <= 2e-4 * max(1, abs(a), abs(b))) else "   <<<<"
        print("  %-6s G % .8e  C % .8e%s" % (n, a, b, flag))
