import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np
from ky_amd import _abi as A, api
from oracle import kyoracle as O
def rmse(a,b): return float(np.sqrt(np.mean((np.asarray(a,np.float64)-np.asarray(b,np.float64))**2)))
for (w, h, spp, depth) in ((1, 1, 1, 5), (13, 7, 3, 5), (13, 7, 64, 0), (13, 7, 16, 250)):
    sc = api.cornell_box_scene(A.CB_DEFAULT_SCENE, w, h); p = api.make_params(w, h, spp, max_path_depth=depth)
    print("edge", w, h, spp, depth, "%.3e" % rmse(api.render(sc, p), O.render(sc, p)))
for case in ["cornell_area", "cornell_env", "cornell_point", "cornell_direction", "veach", "cornell_depth16", "direct_lighting"]:
    kw = {}
    if case != "veach":
        flag = {"cornell_area": A.CB_LIGHT_AREA, "cornell_env": A.CB_LIGHT_ENVIRONMENT, "cornell_point": A.CB_LIGHT_POINT, "cornell_direction": A.CB_LIGHT_DIRECTION, "cornell_depth16": A.CB_LIGHT_AREA, "direct_lighting": A.CB_LIGHT_AREA}[case]
        W, H = 48, 40; scene = api.cornell_box_scene(A.CB_BOTH_SMALL_SPHERES | flag, W, H)
        if case == "cornell_depth16": kw["max_path_depth"] = 16
        if case == "direct_lighting": kw["integrator"] = A.INTEGRATOR_DIRECT_LIGHTING
    else:
        W, H = 64, 36; scene = api.mis_scene(W, H)
    p = api.make_params(W, H, 1024, tile_w=16, tile_h=8, **kw)
    g, c = api.render(scene, p), O.render(scene, p)
    fin = np.isfinite(c).all(axis=2)
    print("film", case, "%.3e" % rmse(g[fin], c[fin]), "nonfinite", int((~fin).sum()))
# smoke
scene = api.cornell_box_scene(A.CB_DEFAULT_SCENE, 64, 64); params = api.make_params(64, 64, 32)
print("smoke %.3e" % rmse(api.render(scene, params), O.render(scene, params)))
# per-sample all light flags/strategies
import test_parity_gpu as T
for flag in ("area","direction","point","environment"):
    f = {"area": A.CB_LIGHT_AREA, "direction": A.CB_LIGHT_DIRECTION, "point": A.CB_LIGHT_POINT, "environment": A.CB_LIGHT_ENVIRONMENT}[flag]
    scene = api.cornell_box_scene(A.CB_BOTH_SMALL_SPHERES | f, 64, 64)
    for strat in T.STRATEGIES:
        params = api.make_params(64, 64, 128, direct_sample=strat)
        pixels = [(32, 32), (5, 5), (21, 42), (44, 45), (60, 61), (32, 4), (18, 50), (46, 52)]
        bad, tot, sg, sc = T.li_agreement(api, O, scene, params, pixels)
        print("li cornell", flag, strat, bad, tot, "%.2e" % (abs(sg-sc)/max(sc,1)))
scene = api.mis_scene(96, 54)
for depth in (5,16):
    for strat in T.STRATEGIES:
        params = api.make_params(96, 54, 128, direct_sample=strat, max_path_depth=depth)
        pixels = [(48, 27), (5, 5), (30, 40), (70, 30), (48, 50), (20, 20), (80, 45), (60, 8)]
        bad, tot, sg, sc = T.li_agreement(api, O, scene, params, pixels)
        print("li veach", depth, strat, bad, tot, "%.2e" % (abs(sg-sc)/max(sc,1)))
