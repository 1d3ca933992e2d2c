#!/usr/bin/env python3
"""Work per camera sample of every frame bench.py renders, counted by the CPU oracle (oracle/ky_oracle.cpp's counters: the same events
SURVEY.md section 6 counted on the reference with gprof -- traversals, primitive tests, NEE vertices, light estimates, continuation
samples, path iterations, MIS rays, roulette draws) -> tests/golden/work_counters.json.

These are the algorithm's counts, independent of how the GPU kernel is built; bench.py's `roofline.valu_model` prices them with the
cheapest instruction sequences the scene's shapes admit to get the USEFUL lane-instructions per sample.
tests/test_oracle_pins.py holds the oracle's counters of configs[1] / [2] to the reference's own within 1.2 %.

usage: python tests/golden/make_work_counters.py     (CPU only, about a minute)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ky_amd import _abi as A, api   # noqa: E402  (host-side scene builders only: no GPU is touched)
from oracle import kyoracle as O     # noqa: E402

FRAMES = {   # label -> (scene, width, height, spp, depth): the frames of bench.py's workloads at reduced size, same aspect
    "cornell": (lambda w, h: api.cornell_box_scene(A.CB_DEFAULT_SCENE, w, h), 256, 192, 64, 5),
    "veach": (lambda w, h: api.mis_scene(w, h), 320, 180, 64, 5),
    "cornell_d16": (lambda w, h: api.cornell_box_scene(A.CB_DEFAULT_SCENE, w, h), 192, 192, 64, 16),
    "cornell_point": (lambda w, h: api.cornell_box_scene(A.CB_BOTH_SMALL_SPHERES | A.CB_LIGHT_POINT, w, h), 192, 192, 64, 5),
    "cornell_direction": (lambda w, h: api.cornell_box_scene(A.CB_BOTH_SMALL_SPHERES | A.CB_LIGHT_DIRECTION, w, h), 192, 192, 64, 5),
    "cornell_area": (lambda w, h: api.cornell_box_scene(A.CB_BOTH_SMALL_SPHERES | A.CB_LIGHT_AREA, w, h), 192, 192, 64, 5),
    "cornell_environment": (lambda w, h: api.cornell_box_scene(A.CB_BOTH_SMALL_SPHERES | A.CB_LIGHT_ENVIRONMENT, w, h), 192, 192, 64, 5),
    "veach_square": (lambda w, h: api.mis_scene(w, h), 192, 192, 64, 5),
}


def main():
    out = {"_comment": "per camera sample, CPU oracle (tests/golden/make_work_counters.py); path_tracing_iteration_t, both_mis, random sampler seed 1234"}
    for label, (make, w, h, spp, depth) in FRAMES.items():
        scene = make(w, h)
        _, c = O.render(scene, api.make_params(w, h, spp, max_path_depth=depth), counters=True)
        n = c["camera_samples"]
        row = {k: round(v / n, 4) for k, v in c.items() if k != "camera_samples"}
        row.update(width=w, height=h, spp=spp, max_path_depth=depth, surfaces=int(scene.c.surface_count), lights=int(scene.c.light_count))
        out[label] = row
        print(label, row)
    with open(os.path.join(ROOT, "tests", "golden", "work_counters.json"), "w") as fh:
        json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
