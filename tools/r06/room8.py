import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
os.environ.setdefault("KYHIP_JIT", "0")
import numpy as np
from ky_amd import _abi as A, api
from oracle import kyoracle as O
from test_random_scenes_gpu import random_room
lib = A.load_kyhip()
W, H = 48, 40
for seed in (4250,):
    scene, kinds = random_room(A, api, O, seed, False, W, H)
    print("room", seed, kinds, "facts", api.scene_facts(scene))
    for strategy in (A.DIRECT_BOTH_MIS, A.DIRECT_LIGHT_MIS):
        p = api.make_params(W, H, 64, direct_sample=strategy, tile_w=16, tile_h=8)
        inline = api.render(scene, p); ik = lib.kyhip_last_kernel(0)
        prev = lib.kyhip_set_shadow_queue(1)
        deferred = api.render(scene, p); dk = lib.kyhip_last_kernel(0)
        lib.kyhip_set_shadow_queue(prev)
        d = np.abs(inline - deferred).max(axis=2)
        print(strategy, ik, dk, d.max(), (d > 2e-6).sum(), np.argwhere(d > 2e-6)[:5].tolist())
        lib.kyhip_set_specialisation(0)
        gen = api.render(scene, p); gk = lib.kyhip_last_kernel(0)
        lib.kyhip_set_specialisation(1)
        print("   generic", gk, np.abs(inline - gen).max(), np.abs(deferred - gen).max())
