"""lane engine vs queue engine vs oracle on small frames (run on the GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ky_amd import api, _abi as A
from oracle import kyoracle as O
lib = A.load_kyhip()
for name, scene in (("cornell", api.cornell_box_scene(A.CB_DEFAULT_SCENE, 96, 72)), ("veach", api.mis_scene(96, 72))):
    for spp in (1, 4, 40, 200):
        p = api.make_params(96, 72, spp)
        lib.kyhip_set_engine(0); a = api.render(scene, p)
        lib.kyhip_set_engine(1); b = api.render(scene, p); b2 = api.render(scene, p)
        c = O.render(scene, p)
        d = np.abs(a - b)
        print(name, "spp", spp, "lane-vs-queue max %.3e" % d.max(), "px>1e-5:", int((d.max(axis=2) > 1e-5).sum()), "queue repeatable:", bool(np.array_equal(b, b2)),
              "means %.7f %.7f %.7f" % (a.mean(), b.mean(), c.mean()),
              "rmse lane/oracle %.2e queue/oracle %.2e" % (np.sqrt(np.nanmean((a - c) ** 2)), np.sqrt(np.nanmean((b - c) ** 2))), flush=True)
