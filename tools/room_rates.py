"""The random rooms of tests/test_random_scenes_gpu.py as a rate measurement: every room at 512x384, 64 spp, both_mis, on the kernel the library's table
gives it and on the kernel compiled for its exact template arguments (kyhip_set_jit).  What a scene nobody tuned for loses to the table.
usage: tools/room_rates.py [rooms, default 72 | a,b,c] [spp, default 64]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from ky_amd import api, _abi as A
from test_random_scenes_gpu import random_room


class Bounds:   # the rooms' world radius without the oracle (a rate does not care about the last digit of it)
    @staticmethod
    def world_bounding_sphere(scene):
        return (0.0, 0.0, 0.0, 2.3)


lib = A.load_kyhip()
arg = sys.argv[1] if len(sys.argv) > 1 else "72"
seeds = [int(x) for x in arg.split(",")] if "," in arg else list(range(int(arg)))
W, H, SPP = 512, 384, int(sys.argv[2]) if len(sys.argv) > 2 else 64
rows = []
for seed in seeds:
    scene, kinds = random_room(A, api, Bounds, 4242 + seed, seed % 2 == 1, W, H)
    p = api.make_params(W, H, SPP)
    out = []
    for jit in (0, 1):
        lib.kyhip_set_jit(jit)
        api.render(scene, p)
        best = 1e9
        for _ in range(3):
            api.render(scene, p); best = min(best, api.kernel_ms())
        out.append((lib.kyhip_last_kernel(0).decode(), best))
    lib.kyhip_set_jit(0)
    rows.append((seed, "+".join(kinds) + (" (general)" if seed % 2 else ""), out))
    print("room %2d %-44s table %-58s %7.3f ms | own %-62s %7.3f ms | table / own %.3f" % (seed, rows[-1][1], out[0][0], out[0][1], out[1][0], out[1][1], out[0][1] / out[1][1]), flush=True)
r = np.array([o[0][1] / o[1][1] for _, _, o in rows])
same = sum(1 for _, _, o in rows if o[0][0] == o[1][0])
print("# %d rooms: table / own kernel time  min %.3f  median %.3f  max %.3f; %d rooms on a table row that IS their exact instantiation; rooms whose table kernel is slower than 1 / 0.9 of their own: %d" % (
    len(rows), r.min(), np.median(r), r.max(), same, int((r > 1 / 0.9).sum())))
by = {}
for _, _, o in rows:
    by.setdefault(o[0][0], []).append(o[0][1] / o[1][1])
for k, v in sorted(by.items(), key=lambda kv: -len(kv[1])):
    print("# table kernel %-60s %2d rooms, table / own %.3f .. %.3f" % (k, len(v), min(v), max(v)))
