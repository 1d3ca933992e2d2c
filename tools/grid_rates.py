#!/usr/bin/env python3
"""Kernel rate of every cell of ky's experiment grids on one GPU: render_multiple_integrator (ky.cpp:4740-4777: 5 integrators x 4 Cornell light
variants, both_mis), render_direct_sample_enum (4779: 5 strategies x 4 lights), render_mis_scene (4878: 6 strategies, Veach) -- at
1024 x 768 (Veach 1280 x 720), `spp` samples.  Prints Gsamples/s from the library's kernel timing and the instantiation that ran."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ky_amd import api, _abi as A
lib = A.load_kyhip()
lib.kyhip_set_jit(0)   # the rates of the TABLE's kernels (run-time instantiations are on by default since round 6)
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 256
lights = (("point", A.CB_LIGHT_POINT), ("direction", A.CB_LIGHT_DIRECTION), ("area", A.CB_LIGHT_AREA), ("environment", A.CB_LIGHT_ENVIRONMENT))
integrators = (("direct_lighting", 6), ("simple_recursion", 8), ("recursion", 9), ("recursion_defered", 10), ("iteration", 11))
strategies = (("idle", 0), ("bsdf", 4), ("light", 8), ("bsdf_mis", 16), ("light_mis", 32), ("both_mis", 48))
W, H = 1024, 768


def rate(scene, p):
    api.render(scene, p)
    best = 1e9
    for _ in range(2):
        api.render(scene, p)
        best = min(best, api.kernel_ms())
    return p.width * p.height * p.samples_per_pixel / best / 1e6, lib.kyhip_last_kernel(0).decode()


print("== render_multiple_integrator: integrator x light, both_mis, depth 5, %d spp (Gsamples/s)" % spp)
for iname, integ in integrators:
    row = []
    for lname, flag in lights:
        g, k = rate(api.cornell_box_scene(A.CB_BOTH_SMALL_SPHERES | flag, W, H), api.make_params(W, H, spp, integrator=integ))
        row.append("%s %6.2f" % (lname, g))
    print("%-18s %s   [%s]" % (iname, "  ".join(row), k))
print("== render_direct_sample_enum: strategy x light, iteration")
for sname, st in strategies:
    row = []
    for lname, flag in lights:
        g, k = rate(api.cornell_box_scene(A.CB_BOTH_SMALL_SPHERES | flag, W, H), api.make_params(W, H, spp, direct_sample=st))
        row.append("%s %6.2f" % (lname, g))
    print("%-18s %s   [%s]" % (sname, "  ".join(row), k))
print("== render_mis_scene: Veach 1280x720, iteration")
veach = api.mis_scene(1280, 720)
for sname, st in strategies:
    g, k = rate(veach, api.make_params(1280, 720, spp, direct_sample=st))
    print("%-18s %6.2f   [%s]" % (sname, g, k))
