#!/usr/bin/env python3
"""bench.py -- headline metric of BASELINE.json on its named configurations.

  metric    Msamples/s (paths x spp): camera samples per wall second, whole job.
  workloads --workload cornell  configs[1] (default): ky Cornell box (both_small_spheres | light_area), 1024x768, 1024 spp,
                                path_tracing_iteration depth 5, both_mis
            --workload veach    configs[2]: Veach MIS scene 1280x720, 4096 spp, same integrator
            --workload batch    configs[3]: the render_multiple_scene batch -- four Cornell light variants + Veach (both_mis) +
                                a first-hit AOV pass, each 1024x1024 at 2048 spp -- six frames per step
            --workload stress   configs[4]: Cornell 4096x4096, 16384 spp, max depth 16 (2.7e11 samples per step: pass --steps 1
                                --warmup 0, or scale with --spp)
            --workload single   ky's own default driver, render_single_scene (ky.cpp:4675-4712): the Cornell box lit by the ENVIRONMENT light,
                                1024x1024, path_tracing_iteration depth 5 both_mis -- at 2048 spp instead of the driver's 16 (the rate does not
                                depend on spp from 256 up)
            One "step" renders the workload's frame(s) once.
  N > 1     every frame's tiles are interleaved over the N ranks (one process per GPU), each rank renders its tiles with no
            communication, then ONE gather of film tiles to rank 0 (RCCL over xGMI) and one add kernel there.  Total work is
            fixed, so scaling is "strong".

Launch: `python bench.py` (N=1) or
        `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
             bench.py --gpus N --steps K --warmup W`.
Rank 0 prints ONE JSON line.  The CPU oracle is used here ONLY for the reported `cpu_baseline` (and the RMSE next to it); it is
never inside the timed GPU region.

Beside the contract's fields the default N = 1 line carries (all measured live, outside the timed region):
  extra_workloads     one step each of configs[2] (veach), configs[3] (batch) and configs[4] (stress) with their kernel durations
  projected_scaling   kernel-level strong-scaling efficiency for N = 2, 4, 8 from rendering every 1/N shard of the frame on this
                      one GPU (the slowest shard sets the pace), and the cost of an N = 8 shard when launches are pipelined
  roofline.valu       the counters of the real bound (VALU issue) from profiles/valu.json, flagged stale when the live kernel time has
                      moved away from the profiled one
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

# the CPU baseline runs with its threads pinned to cores.  An OpenMP runtime reads these when it starts, and
# numpy / torch bring one with them: set them before anything is imported.
os.environ.setdefault("OMP_PROC_BIND", "close")
os.environ.setdefault("OMP_PLACES", "cores")


def _quota_cpus():   # (cpus_granted() below, needed here before the imports: OpenMP sizes its teams by the affinity mask, not by the cgroup's quota)
    granted = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            granted = min(granted, max(1, int(round(int(quota) / int(period)))))
    except Exception:
        pass
    return granted


os.environ.setdefault("OMP_NUM_THREADS", str(_quota_cpus()))

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from ky_amd import _abi as A  # noqa: E402
from ky_amd import api, dist as kydist  # noqa: E402

JIT_DEFAULT_MODE = None   # kyhip_set_jit's mode when this process started (main() pins 0 for the timed workloads)
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s
MAX_CLOCK_HZ = 2.4e9   # the same guide: max clock 2400 MHz; peak FP32 (vector) 157.3 TFLOP/s = 256 CUs x 128 lanes x 2 (FMA) x 2.4 GHz


def bytes_per_sample(iterations):
    """algorithmic bytes per camera sample of an HBM ray-pool tracer = 128 B x mean path iterations + 12 B film."""
    return 128.0 * iterations + 12.0


# mean path iterations per camera sample (scene->intersect calls made by Li): SURVEY.md section 6 for configs[1] / [2]
# (4.168 / 2.711, measured on the reference); the other frames with the oracle's counters (oracle/ky_oracle.cpp, 128x128x64):
# Cornell at depth 16: 4.266 (roulette ends paths long before the cap); point / direction / environment lights: 4.215;
# Veach at 1:1 aspect: 2.803; a first-hit AOV pass: 1.
ITER = {"cornell": 4.168, "veach": 2.711, "cornell_d16": 4.266, "cornell_other_lights": 4.215, "veach_square": 2.803, "aov": 1.0}


# ---- roofline.valu_model: USEFUL lane-instructions per camera sample -- a floor, never more than what ran ------------------------------
# The path is bound by VALU issue, so its roofline is lane-slots: CUs x 4 SIMDs x 32 fp32 lanes per clock at the maximum
# engine clock (the guide's 157.3 TFLOP/s of vector fp32 FMAs is the same figure x 2).  What fills a slot usefully is one fp32 / int32 VALU
# operation of ONE path that the CHEAPEST ALGORITHM for the reference's result needs: the reference's event counts per camera sample
# (tests/golden/work_counters.json: the oracle's counters of each frame; tests/test_oracle_pins.py holds them to SURVEY section 6's gprof
# counts of the reference within 1.2 %), each priced with the least work that decides it (round 5; rounds 3-4 priced every ray as the
# reference traces it -- every surface of the scene -- and so credited work the kernel legitimately skips: 0.82 on configs[2] for a
# kernel whose counters say 0.50):
#   nearest-hit ray      every surface once (no scan of a dozen surfaces beats the branch-free linear one: DESIGN.md 10)
#   MIS BSDF ray         "is the nearest hit a carrier of this light" (3989-3995): the carrier's own test -- one parallelogram (a Cornell lamp)
#                        or the discriminant half of a sphere test (a Veach lamp: a line that misses the sphere needs no root)
#   shadow ray           occluded: ONE surface test (the first surface tried is the blocker at best); unoccluded: the light's occluder table
#                        once (DScene::occ_front / occ / trav: the surfaces no proof removes)
# so that useful <= executed by construction: main() checks it against the kernel's own counters whenever profiles/valu.json describes the
# loaded kernels (`valu_model_exceeds_executed`), and reports the executed fraction (`lane_slot_frac`) beside `frac`.
LANE_OP = {
    "aar_test": 12,        # rectangle in an axis plane, incl. the nearest / any update: sub, mul, 2 x (fma, sub), 4 compares, 2 moves
    "box_test": 40,        # up to six rectangles that are whole faces of one axis-aligned box (round 5, DESIGN.md 3 "boxes"): 6 x (sub, mul), 6 tags, 3 min, 3 max, max3, min3, 2 x (and, 4 compares, 2 moves)
    "par_test": 26,        # planar parallelogram: 2 dots (6), rcp, mul, hit point (3), 2 dual-basis dots + offsets (8), 4 compares, 2 moves
    "sphere_test": 21,     # oc (3), b (3), |oc|^2 (3), discriminant (2), sqrt, 2 roots, 4 compares, select, 2 selects
    "sphere_reject": 11,   # the same up to the discriminant's sign (a lamp that subtends a thousandth of the directions: sph_hit's `sparse` form)
    "traversal_setup": 7,  # 3 reciprocals of the direction, tmax / best initialisation
    "vertex": 32,          # hit point 3, normal fetch + flip / normalise 8, emission test 5, termination tests 3, material + lobe 6, ray spawn 7
    "frame": 12,           # frame_t(n) for a non-delta vertex (566-571): rsq, 2 mul, cross
    "light_rect": 22, "light_sphere": 45, "light_point": 12, "light_direction": 7, "light_environment": 24,   # light_t::sample_Li incl. pdf and the facing test
    "shadow_setup": 12,    # direction, distance - 2e-3, offset origin (3187-3201)
    "bsdf_eval": 14,       # eval_ + pdf_ of the vertex's lobe for the light sample (Lambert 10; a Phong lobe's pow adds ~25 on the lanes that hold one)
    "mis": 6,              # 2 f cos L / (p + q) (4028 / 4070), 0.5 x, accumulate
    "bsdf_dir": 29,        # concentric disk 16, z 4, basis combination 9 (the Phong lobe's mapping costs about the same)
    "ray_spawn": 7,        # offset_ray_origin (614-620)
    "continuation": 46,    # bsdf_dir + weight 6 + ray_spawn + roulette 4
    "rng_draw": 8,         # xoroshiro64+ (add, xor, 2 rotates, shift, three-way xor) incl. the two-instruction conversion to [0, 1)
    "film": 4,             # Lo / spp into the pixel's sum
}
# per frame: the whole scene (rectangles in an axis plane that are no box's face, other parallelograms, spheres, boxes: the Cornell room's five walls are one box,
# the lamp housing's five rectangles another -- 40 lane-instructions instead of 5 x 12, so the floor prices the box), the light kind of each light estimate, what an MIS BSDF ray must
# test (LANE_OP key; None: the light has no BSDF-sampling half or no carrier surface), and the occluder table an UNOCCLUDED shadow ray scans (same triple;
# spheres of a sphere-lights scene at the reject price).  The Cornell lamp's table is DScene::occ_front (the lamp's rectangle + the two balls); the delta
# lights' is DScene::occ (the room's five walls proved away: two balls); the environment light's rays leave the room: every surface; Veach: no wall
# qualifies (its floor reaches under the back wall), every surface.
SCENE_SHAPES = {
    # (the Cornell lamp is a rectangle in an axis plane: an MIS ray's carrier test is the 12-instruction one since round 5's KY_FEAT_AXIS_ALIGNED)
    "cornell": ((0, 0, 2, 2), "rect", "aar_test", (1, 0, 2)), "cornell_area": ((0, 0, 2, 2), "rect", "aar_test", (1, 0, 2)), "cornell_d16": ((0, 0, 2, 2), "rect", "aar_test", (1, 0, 2)),
    "cornell_point": ((0, 0, 2, 1), "point", None, (0, 0, 2)), "cornell_direction": ((0, 0, 2, 1), "direction", None, (5, 0, 2)),
    "cornell_environment": ((0, 0, 2, 1), "environment", None, (5, 0, 2)),
    "veach": ((2, 4, 5, 0), "sphere", "sphere_reject", (2, 4, 5)), "veach_square": ((2, 4, 5, 0), "sphere", "sphere_reject", (2, 4, 5)),
}


def work_counters():
    try:
        with open(os.path.join(ROOT, "tests", "golden", "work_counters.json")) as fh:
            return json.load(fh)
    except Exception:
        return {}


def useful_lane_ops(label, counters):
    """-> (lane-instructions per camera sample, its terms) for path_tracing_iteration_t with both_mis on the frame `label`."""
    c = counters.get(label)
    if not c or label not in SCENE_SHAPES:
        return None, None
    (n_aar, n_par, n_sph, n_box), light, mis_test, (o_aar, o_par, o_sph) = SCENE_SHAPES[label]
    sphere_lamps = light == "sphere"
    per_nearest = (n_aar * LANE_OP["aar_test"] + n_par * LANE_OP["par_test"] + n_sph * LANE_OP["sphere_test"] + n_box * LANE_OP["box_test"] + LANE_OP["traversal_setup"])
    per_unoccluded = (o_aar * LANE_OP["aar_test"] + o_par * LANE_OP["par_test"] + o_sph * LANE_OP["sphere_reject" if sphere_lamps else "sphere_test"]
                      + LANE_OP["traversal_setup"])
    per_occluded = LANE_OP["sphere_test"] if sphere_lamps else LANE_OP["aar_test"]   # the blocker itself: a lamp's own sphere (quirk 1) / the Cornell lamp's rectangle
    nearest = c["traversals"] - c["shadow_rays"] - c["mis_bsdf_rays"]
    # an environment light's BSDF-sampled ray asks for the NEAREST hit of the whole scene (a miss counts): no carrier to test instead
    per_mis = LANE_OP[mis_test] if mis_test else (per_nearest if light == "environment" else 0)
    area = light in ("rect", "sphere", "environment")   # lights whose estimate has a BSDF-sampling half (a delta light's returns black, 3977)
    per_estimate = LANE_OP["light_" + light] + LANE_OP["shadow_setup"] + LANE_OP["bsdf_eval"] + LANE_OP["mis"] + ((LANE_OP["bsdf_dir"] + LANE_OP["ray_spawn"]) if area else 0)
    draws = 2 + 4 * c["light_estimates"] + 2 * c["bsdf_path_samples"] + c["rr_draws"]
    terms = {
        "nearest_hit_rays": nearest * per_nearest,
        "mis_bsdf_rays": c["mis_bsdf_rays"] * per_mis,
        "shadow_rays": (c["shadow_rays"] - c["shadow_occluded"]) * per_unoccluded + c["shadow_occluded"] * per_occluded,
        "path_vertices": c["path_iterations"] * LANE_OP["vertex"] + c["nee_vertices"] * LANE_OP["frame"],
        "light_estimates": c["light_estimates"] * per_estimate,
        "continuation": c["bsdf_path_samples"] * LANE_OP["continuation"],
        "random_numbers": draws * LANE_OP["rng_draw"],
        "film": LANE_OP["film"],
    }
    return sum(terms.values()), terms


def frame_iterations(label, fallback):
    """mean path iterations per camera sample of the frame `label`: its own counter (tests/golden/work_counters.json) where there is one --
    configs[1]'s 4:3 frame holds 3.489, not the square frame's 4.168 SURVEY section 6 quotes."""
    c = work_counters().get(label)
    return float(c["path_iterations"]) if c and "path_iterations" in c else fallback


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cornell", choices=["cornell", "veach", "batch", "stress", "single"])
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--spp", type=int, default=0)
    ap.add_argument("--depth", type=int, default=0)
    ap.add_argument("--direct-sample", type=int, default=A.DIRECT_BOTH_MIS, choices=[0, 4, 8, 16, 32, 48],
                    help="direct_sample_enum_t of the path integrator frames (profiling the run-time-dispatched kernel; the metric's configs use 48)")
    ap.add_argument("--integrator", type=int, default=A.INTEGRATOR_PATH_TRACING_ITERATION, choices=[6, 8, 9, 10, 11],
                    help="integrator_enum_t of the path frames (profiling direct_lighting_t = 6 and the recursive integrators 8 / 9 / 10; the metric's configs use 11)")
    ap.add_argument("--no-pipeline", action="store_true", help="render the timed steps on one stream (no overlap of a frame's start with the previous frame's tail): `value` is then the single-frame rate")
    ap.add_argument("--pipeline", action="store_true", help="(the default since round 5, at every N: consecutive frames on two streams; kept so that old command lines still parse)")
    ap.add_argument("--no-extra", action="store_true", help="skip extra_workloads and projected_scaling (they run for the default N = 1 cornell line only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the cpu_baseline sample")
    return ap.parse_args()


class Frame:
    def __init__(self, label, scene, params, iterations, origin=(0, 0)):
        self.label, self.scene, self.params, self.origin = label, scene, params, origin
        self.iterations = frame_iterations(label, iterations)   # the frame's own count where the oracle's counters hold one, SURVEY's otherwise
        self.bytes_per_sample = bytes_per_sample(self.iterations)
        self.samples = params.width * params.height * params.samples_per_pixel


def workload(args):
    """-> (frames, film (height, width), name)"""
    if args.workload == "cornell":
        W, H, spp, depth = args.width or 1024, args.height or 768, args.spp or 1024, args.depth or 5
        frames = [Frame("cornell", api.cornell_box_scene(A.CB_DEFAULT_SCENE, W, H), api.make_params(W, H, spp, max_path_depth=depth, direct_sample=args.direct_sample, integrator=args.integrator), ITER["cornell"])]
        name = "BASELINE configs[1]: ky Cornell box (both_small_spheres|light_area) %dx%d, %d spp, path_tracing_iteration d%d both_mis" % (W, H, spp, depth)
        return frames, (H, W), name
    if args.workload == "veach":
        W, H, spp, depth = args.width or 1280, args.height or 720, args.spp or 4096, args.depth or 5
        frames = [Frame("veach", api.mis_scene(W, H), api.make_params(W, H, spp, max_path_depth=depth, direct_sample=args.direct_sample, integrator=args.integrator), ITER["veach"])]
        name = "BASELINE configs[2]: ky Veach MIS scene (create_mis_scene) %dx%d, %d spp, path_tracing_iteration d%d both_mis" % (W, H, spp, depth)
        return frames, (H, W), name
    if args.workload == "stress":
        W, H, spp, depth = args.width or 4096, args.height or 4096, args.spp or 16384, args.depth or 16
        frames = [Frame("cornell_d16", api.cornell_box_scene(A.CB_DEFAULT_SCENE, W, H), api.make_params(W, H, spp, max_path_depth=depth, direct_sample=args.direct_sample, integrator=args.integrator), ITER["cornell_d16"])]
        name = "BASELINE configs[4]: stress, ky Cornell box %dx%d, %d spp, path_tracing_iteration d%d both_mis" % (W, H, spp, depth)
        return frames, (H, W), name
    if args.workload == "single":
        W, H, spp, depth = args.width or 1024, args.height or 1024, args.spp or 2048, args.depth or 5
        frames = [Frame("cornell_environment", api.cornell_box_scene(A.CB_BOTH_SMALL_SPHERES | A.CB_LIGHT_ENVIRONMENT, W, H),
                        api.make_params(W, H, spp, max_path_depth=depth, direct_sample=args.direct_sample, integrator=args.integrator), ITER["cornell_other_lights"])]
        name = "ky render_single_scene (ky.cpp:4675-4712): Cornell box (both_small_spheres|light_environment) %dx%d, %d spp, path_tracing_iteration d%d both_mis" % (W, H, spp, depth)
        return frames, (H, W), name
    # batch: render_multiple_scene (ky.cpp:4819-4876) at production size, one film_grid_t(2, 3, res, res)
    res, spp, depth = args.width or 1024, args.spp or 2048, args.depth or 5
    frames = []
    lights = (("point", A.CB_LIGHT_POINT), ("direction", A.CB_LIGHT_DIRECTION), ("area", A.CB_LIGHT_AREA), ("environment", A.CB_LIGHT_ENVIRONMENT))
    for cell, (lname, flag) in enumerate(lights):
        it = ITER["cornell"] if lname == "area" else ITER["cornell_other_lights"]
        frames.append(Frame("cornell_" + lname, api.cornell_box_scene(A.CB_BOTH_SMALL_SPHERES | flag, res, res), api.make_params(res, res, spp, max_path_depth=depth, direct_sample=args.direct_sample, integrator=args.integrator),
                            it, ((cell % 3) * res, (cell // 3) * res)))
    veach = api.mis_scene(res, res)
    frames.append(Frame("veach_square", veach, api.make_params(res, res, spp, max_path_depth=depth, direct_sample=args.direct_sample, integrator=args.integrator), ITER["veach_square"], (res, res)))
    frames.append(Frame("veach_normal_aov", veach, api.make_params(res, res, 1, integrator=A.INTEGRATOR_NORMAL, sampler=A.SAMPLER_DEBUG), ITER["aov"], (2 * res, res)))
    name = ("BASELINE configs[3]: render_multiple_scene batch, 4 Cornell light variants + Veach (path_tracing_iteration d%d both_mis) + first-hit AOV, "
            "each %dx%d at %d spp, film_grid 2x3" % (depth, res, res, spp))
    return frames, (2 * res, 3 * res), name


REFERENCE_CPU = {   # the reference itself (unmodified ky.cpp, g++ -O3 -fopenmp) in the survey container, 8 cores
    "cornell": {"Msamples/s": 1.374, "cores": 8, "what": "ky.cpp Cornell 256x256x16, path_tracing_iteration d5 both_mis, 8-core Xeon 2.1 GHz VM"},
    "veach": {"Msamples/s": 0.609, "cores": 8, "what": "ky.cpp Veach 320x180x16, same integrator, 8-core Xeon 2.1 GHz VM"},
}


def cpus_granted():
    """What the host really grants this process: the affinity mask, cut down to the cgroup's CPU quota where there is one."""
    granted = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            granted = min(granted, max(1, int(round(int(quota) / int(period)))))
    except Exception:
        pass
    return granted


def film_pixels_of(p):
    """Pixels INSIDE the film of the tiles (tile_first, tile_step) of p (edge tiles are clipped)."""
    n = 0
    for t in range(p.tile_first, kydist.tiles_total(p), p.tile_step):
        x0, y0 = kydist.tile_origin(p, t)
        n += max(0, min(p.tile_w, p.width - x0)) * max(0, min(p.tile_h, p.height - y0))
    return n


def rmse_of(pairs):
    """RMSE over the lit, finite pixels of (gpu_film, cpu_film) pairs.  The reference's own arithmetic yields inf * 0 = NaN for a few
    exactly-grazing mirror hits (DESIGN.md "Non-finite samples"); such pixels are excluded and counted."""
    se, cnt, bad = 0.0, 0, 0
    for gpu_film, cpu_film in pairs:
        lit = (cpu_film != 0).any(axis=2) | (gpu_film != 0).any(axis=2)
        fin = np.isfinite(cpu_film).all(axis=2) & np.isfinite(gpu_film).all(axis=2)
        d = gpu_film[fin & lit].astype(np.float64) - cpu_film[fin & lit].astype(np.float64)
        se += float((d * d).sum()); cnt += d.size; bad += int((~fin).sum())
    return (se / max(cnt, 1)) ** 0.5, bad


def sample_params(fr, spp, pixels=131072):
    """fr's parameters cut down to an interleaved subset of its tiles (about `pixels` pixels spread over the whole picture) at `spp` samples per pixel."""
    p = A.RenderParams.from_buffer_copy(fr.params)
    p.tile_first = 0
    p.tile_step = max(1, kydist.tiles_total(fr.params) * fr.params.tile_w * fr.params.tile_h // pixels)
    p.samples_per_pixel = max(1, min(fr.params.samples_per_pixel, spp))
    return p


def parity_full_spp(frames, gpu_render, O, threads, budget_samples=None, min_pixels=0, scene_device=0):
    """RMSE of the GPU's film against the CPU oracle's at the workload's FULL spp on an interleaved tile sample of every path frame -- the figure the north
    star's 1e-3 is about.  The sample is `min_pixels` pixels over the frames, or what `budget_samples` camera samples buy if that is more.  Pixels that are
    off by more than 5e-3 are replayed sample by sample on both sides (kyhip_kat_li / the oracle's li) and classified with the tests' own rules
    (tests/helpers.py, explain_sample): "explained" = every differing sample differs first in a recorded decision or at / after a vertex that amplifies rounding."""
    pairs, kept, full_px, t_full, flips = [], [], 0, 0.0, {"pixels_off_by_5e-3": 0, "explained": 0, "unexplained": 0, "not_examined": 0}
    path_frames = [fr for fr in frames if fr.params.samples_per_pixel > 1]
    per_frame_px = max(1, int(np.ceil(min_pixels / max(len(path_frames), 1))))
    for fr in path_frames:
        px = max(fr.params.tile_w * fr.params.tile_h, per_frame_px, int((budget_samples or 0) / len(path_frames) / fr.params.samples_per_pixel))
        p = sample_params(fr, fr.params.samples_per_pixel, pixels=px)
        t0 = time.perf_counter()
        cpu_film = O.render(fr.scene, p, threads=threads)
        t_full += time.perf_counter() - t0
        gpu_film = gpu_render(fr.scene, p)
        pairs.append((gpu_film, cpu_film))
        full_px += film_pixels_of(p)
        fin = np.isfinite(cpu_film).all(axis=2) & np.isfinite(gpu_film).all(axis=2)
        off = np.where(fin, np.abs(gpu_film.astype(np.float64) - cpu_film.astype(np.float64)).max(axis=2), 0.0)
        ys, xs = np.nonzero(off > 5e-3)
        flips["pixels_off_by_5e-3"] += int(len(ys))
        order = np.argsort(-off[ys, xs])
        keep = np.ones(off.shape, bool)
        for i in order[:24]:   # the two dozen worst of a frame are replayed (a replay is spp samples on both sides: milliseconds)
            try:
                from tests import helpers as TH
                kinds = TH.explain_pixel(api, O, fr.scene, p, int(xs[i]), int(ys[i]), value_tol=2e-3 if fr.label.startswith("veach") else 2e-4,
                                         geom_tol=1e-3 if fr.label.startswith("veach") else 1e-4)
                if sum(kinds.values()) > 0:
                    flips["explained"] += 1
                    keep[ys[i], xs[i]] = False
                else:
                    flips["unexplained"] += 1
                    flips.setdefault("review_pixels", []).append([fr.label, int(xs[i]), int(ys[i]), float(off[ys[i], xs[i]]), "no differing sample found by the replay"])
            except AssertionError as e:
                flips["unexplained"] += 1
                flips.setdefault("review_pixels", []).append([fr.label, int(xs[i]), int(ys[i]), float(off[ys[i], xs[i]]), str(e)[:300]])
            except Exception as e:
                flips["not_examined"] += 1
                flips.setdefault("review_pixels", []).append([fr.label, int(xs[i]), int(ys[i]), float(off[ys[i], xs[i]]), "replay failed: " + repr(e)[:200]])
        flips["not_examined"] += max(0, int(len(ys)) - 24)
        kept.append((np.where(keep[..., None], gpu_film, cpu_film), cpu_film))   # the explained pixels set aside (their difference zeroed)
    rmse_full, bad_full = rmse_of(pairs)
    rmse_kept, _ = rmse_of(kept)
    # `value`: every finite pixel.  `value_without_explained`: without the pixels whose every differing sample is explained -- one camera sample in ~1e6 takes another
    # discrete decision than the oracle's (a hit at its epsilon, a shadow ray at its threshold: tests/test_mismatch_gpu.py) and carries up to the lamp's radiance / spp into
    # its pixel; such pixels are 1e-3 of a frame and ARE most of `value` (5.6e-4 with seven of them in 8448 pixels).  The gate is on the second figure and on `value` staying
    # under 2e-3; unexplained or unexamined pixels are listed for review (their differences are inside both figures).
    return {"value": rmse_full, "value_without_explained": rmse_kept, "spp": max(fr.params.samples_per_pixel for fr in frames), "pixels": full_px,
            "excluded_nonfinite_pixels": bad_full, "cpu_seconds": t_full, "target": 1e-3, **flips}


def cpu_baseline(frames, gpu_render, target_seconds, workload_name):
    """Time the CPU oracle (a port of the reference's algorithm; one OpenMP thread per granted CPU, pinned: OMP_PROC_BIND / OMP_PLACES
    are set at the top of this file) on a BOUNDED sample of the same workload: of every frame an interleaved subset of its tiles
    (about 131k pixels spread over the whole picture; the oracle honours tile_first / tile_step) at reduced spp -- the rate depends
    on neither.  Returns the baseline object and the RMSE figures of the GPU's render of the very same samples against it:
    the throughput sample's (reduced spp) and one at the workload's FULL spp on a few tiles, the figure the north star's 1e-3 is about."""
    from oracle import kyoracle as O
    granted = cpus_granted()
    threads = max(1, min(O.max_threads(), granted))

    def run(spp_scale):
        films, n, t = [], 0, 0.0
        for fr in frames:
            p = sample_params(fr, int(round(fr.params.samples_per_pixel * spp_scale)))
            t0 = time.perf_counter()
            film = O.render(fr.scene, p, threads=threads)
            t += time.perf_counter() - t0
            n += film_pixels_of(p) * p.samples_per_pixel
            films.append((p, film))
        return films, n, t

    full_spp = max(fr.params.samples_per_pixel for fr in frames)
    run(1.0 / full_spp)                                   # warm-up at 1 spp: thread pool, page faults
    scale = min(1.0, 8.0 / full_spp)
    films, n, dt = run(scale)
    for _ in range(3):                                    # grow the sample until it is worth about target_seconds of CPU work
        if dt >= 0.6 * target_seconds or scale >= 1.0:
            break
        scale = min(1.0, scale * max(1.5, target_seconds / max(dt, 1e-3)))
        films, n, dt = run(scale)
    rate = n / dt / 1e6
    rmse, bad = rmse_of([(gpu_render(fr.scene, p), cpu_film) for fr, (p, cpu_film) in zip(frames, films)])
    spps = sorted({p.samples_per_pixel for p, _ in films})
    # the same comparison at the workload's full spp: a few tiles per path frame, sized to about a third of the throughput sample's CPU time
    budget = max(2e5, 0.35 * target_seconds * rate * 1e6)
    full = parity_full_spp(frames, gpu_render, O, threads, budget_samples=budget)
    ref = REFERENCE_CPU.get(workload_name)
    cb = {
        "value": rate, "unit": "Msamples/s", "cores": threads, "kind": "port", "omp_threads": threads, "cpus_granted": granted,
        "omp_proc_bind": os.environ.get("OMP_PROC_BIND"), "omp_places": os.environ.get("OMP_PLACES"),
        "sample": "%d frame(s) of the workload, every %d-th tile (interleaved over the picture), %s spp: %.3g samples, %.1f s of CPU work, OpenMP %d threads on %d granted CPUs"
                  % (len(frames), films[0][0].tile_step, "/".join(map(str, spps)), n, dt, threads, granted),
        "reference_itself": ref,   # the reference's own rate where it could be built (other hardware): shows the port is not a sandbagged baseline
    }
    extra = {"rmse_gpu_vs_cpu": rmse, "rmse_spp": spps[-1], "rmse_excluded_nonfinite_pixels": bad, "rmse_full_spp": full}
    return cb, extra


def file_sha256(path):
    import hashlib
    try:
        with open(path, "rb") as fh:
            return hashlib.sha256(fh.read()).hexdigest()
    except Exception:
        return None


def load_json(name):
    try:
        with open(os.path.join(ROOT, "profiles", name)) as fh:
            return json.load(fh)
    except Exception:
        return None


def executed_lane_ops(key, lib_sha):
    """Active lane-instructions per camera sample the kernels of workload `key` really executed: SQ_INSTS_VALU x 64 x lane occupancy from the
    counter set of profiles/valu.json -- None when there is no set for the workload or it was measured on other kernel sources."""
    v = (load_json("valu.json") or {}).get(key)
    if not v or v.get("kernel_source_hash") != lib_sha or not v.get("wave_instr_per_sample") or not v.get("lane_occupancy"):
        return None
    return v["wave_instr_per_sample"] * 64.0 * v["lane_occupancy"]


def contract_frac(frames, kernel_ms_per_frame, world=1):
    """algorithmic bytes of the launch set over the measured kernel time, against the HBM peak."""
    launch_bytes = sum(fr.bytes_per_sample * fr.samples / world for fr in frames)
    achieved = launch_bytes / (sum(kernel_ms_per_frame) * 1e-3) / 1e9
    return achieved, launch_bytes


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: ky_amd has no CPU fallback")
    lib = A.load_kyhip()
    # Run-time instantiations (on by default in a single-process job since round 6) are pinned OFF for everything this file times: every workload of the line runs on the
    # library's table of kernels, whatever a background compile would have switched to and whenever; the `run_time_instantiations` section measures them on its own
    global JIT_DEFAULT_MODE
    JIT_DEFAULT_MODE = int(lib.kyhip_set_jit(-1))
    lib.kyhip_set_jit(0)
    # KY_BENCH_ONE_GPU=1 (testing only): every rank uses cuda:0 and the gather runs over gloo, so that the N > 1 code
    # path can be exercised on a single-GPU box; the numbers of such a run are meaningless.
    one_gpu_test = os.environ.get("KY_BENCH_ONE_GPU") == "1"
    # The device of this rank: LOCAL_RANK among the devices THIS PROCESS sees.  A launcher that masks devices per rank (HIP_VISIBLE_DEVICES /
    # ROCR_VISIBLE_DEVICES = one GPU each, the commonest way to start eight ranks) leaves every rank with device 0 only; one that does not leaves
    # every rank with all of them.  `local_rank % device_count` is right in both cases (and in the one-GPU test, where every rank shares cuda:0).
    n_visible = torch.cuda.device_count()
    rank_on_node = local_rank
    local_rank = local_rank % max(n_visible, 1) if (n_visible == 1 or not one_gpu_test) else 0   # from here on: the HIP ordinal inside this process (torch's and libkyhip's alike)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    comm = {"backend": None, "world": 1}
    if world > 1:
        import torch.distributed as tdist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_gpu_test:
            tdist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            tdist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)   # RCCL over xGMI; executed on hardware in a world of one only (tests/test_rccl_gpu.py)
        comm = {"backend": tdist.get_backend(), "world": tdist.get_world_size()}   # what the communicator itself reports
    # which physical device each rank renders on (rank 0 prints the list: two ranks on one GPU would show here, not in the rate)
    props0 = torch.cuda.get_device_properties(dev)
    me = {"rank": rank, "local_rank": rank_on_node, "hip_ordinal": local_rank, "visible_devices": n_visible,
          "mask": os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES") or os.environ.get("CUDA_VISIBLE_DEVICES"),
          "name": props0.name, "pci_bus_id": getattr(props0, "pci_bus_id", None), "pci_device_id": getattr(props0, "pci_device_id", None),
          "uuid": str(getattr(props0, "uuid", "")) or None}
    ranks = [me]
    if world > 1:
        ranks = [None] * world
        tdist.all_gather_object(ranks, me)

    def barrier():
        if world > 1:
            tdist.barrier()
        torch.cuda.synchronize(dev)

    # Launch overlap: consecutive frames go to two alternating streams, so that the next frame's kernel fills the start-up and the tail of the
    # current frame's persistent kernel (ky_amd/dist.py): 0.3 ms per launch, which is 0.4 % of a frame at N = 1 and 5 % of a 1/8 shard.  Round 5: the
    # SAME mode at every N (rounds 3-4 pipelined only N > 1, which would have credited a scaling curve with overlap N = 1 was denied: VERDICT round 4),
    # and the line carries both rates -- `value` (pipelined, the default) and `single_frame` (every frame on one stream, nothing overlapped).
    # --no-pipeline makes `value` itself the single-frame rate: what the profiling commands use (tools/final_profiles.sh), so that rocprofv3's
    # per-kernel durations are those of kernels that have the chip to themselves.
    pipeline = not args.no_pipeline

    def run_workload(wargs, steps, warmup, record_in_timed=False, pipeline=pipeline):
        """-> dict(frames, name, elapsed, kernel_ms per frame, film_mean): `warmup` untimed and `steps` timed steps of the workload, then an
        untimed pass that reads every frame's kernel duration (HIP events recorded by the library around render_kernel only).
        record_in_timed (the 25-second stress frame): the timed steps themselves read the kernel durations, nothing is rendered twice."""
        frames, (FH, FW_), name = workload(wargs)
        film = torch.zeros((FH, FW_, 3), dtype=torch.float32, device=dev) if rank == 0 else None

        def target(fr):   # the frame's cell of the film (a view: same memory, the film's row stride)
            if film is None:
                return None
            x0, y0 = fr.origin
            return film[y0:y0 + fr.params.height, x0:x0 + fr.params.width]

        kernel_ms = [[] for _ in frames]

        def step(record):
            if film is not None:
                film.zero_()
            for i, fr in enumerate(frames):
                kydist.render_distributed(fr.scene, fr.params, rank, world, local_rank, film=target(fr), pipeline=pipeline and not record)
                if record:
                    torch.cuda.synchronize(dev)
                    kernel_ms[i].append(float(lib.kyhip_kernel_ms(local_rank)))

        # pipelined launches alternate between two side streams, each with its own launch state in the library (work counter, accumulators, events) and its own tile
        # buffers here: with fewer than two warm-up steps the second stream's would be allocated inside the timed region (measured: 45.9 instead of 42.3 ms per step
        # at K = 3, W = 1), so the missing ones are run as untimed priming steps and reported as such
        priming = max(0, 2 - warmup) if pipeline else 0
        for _ in range(warmup + priming):
            step(False)
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            step(record_in_timed)
        barrier()
        elapsed = time.perf_counter() - t0
        if not record_in_timed:
            for _ in range(1 if frames[0].samples > 2e10 else max(1, min(steps, 3))):
                step(True)
            barrier()
        t = torch.tensor([elapsed] + [sum(k) / len(k) for k in kernel_ms], dtype=torch.float64, device=dev)
        if world > 1:
            tdist.all_reduce(t, op=tdist.ReduceOp.MAX)
        return {"frames": frames, "name": name, "elapsed": float(t[0].item()), "kernel_ms": [float(v) for v in t[1:].tolist()], "priming": priming,
                "film_mean": float(film.mean().item()) if film is not None else None}

    R = run_workload(args, args.steps, args.warmup)
    frames, name, elapsed, frame_kernel_ms = R["frames"], R["name"], R["elapsed"], R["kernel_ms"]
    # the same K steps with nothing overlapped (skipped for the 20-second stress frame, where a launch's start-up and tail are 1e-5 of it)
    R1 = None
    if pipeline and frames[0].samples <= 2e10:
        R1 = run_workload(args, args.steps, 1, pipeline=False)

    if rank == 0:
        samples_per_step = sum(fr.samples for fr in frames)
        ms_per_step = elapsed / args.steps * 1e3
        value = samples_per_step * args.steps / elapsed / 1e6
        # roofline of the dominant kernel (render_kernel): one launch per frame; the figure is over the step's launches
        achieved, launch_bytes = contract_frac(frames, frame_kernel_ms, world)
        kernel_total_ms = sum(frame_kernel_ms)
        iteration = args.integrator == A.INTEGRATOR_PATH_TRACING_ITERATION
        both_mis = frames[0].params.direct_sample == A.DIRECT_BOTH_MIS and iteration
        # counters of the real bound: profiles/valu.json and profiles/hbm_traffic.json hold what a builder-run rocprofv3 session measured
        # (tools/final_profiles.sh -> tools/make_valu_json.py).  They are NOT measured by this run: `source` names the files, and `stale` says
        # whether this run's kernel time has moved more than 2 % away from the profiled launch (after scaling to the same sample count).
        vj = load_json("valu.json")
        key = args.workload if both_mis else ("light_mis" if (iteration and args.workload == "cornell" and frames[0].params.direct_sample == A.DIRECT_LIGHT_MIS) else None)
        valu = dict(vj.get(key) or {}) if (vj and key) else None
        if not valu and vj and both_mis and args.workload in ("stress", "batch"):
            valu = dict(vj.get("cornell") or {}, note="counters of the cornell workload (same kernel, same scene family)")
        traffic = None
        if valu:
            per_sample_ms = valu.get("kernel_ms", 0) / max(valu.get("samples_per_launch", 1), 1)
            live_per_sample_ms = kernel_total_ms / (samples_per_step / world)
            same_workload = key == args.workload
            valu["source"] = "profiles/valu.json <- profiles/%s (builder-run rocprofv3 session, not this run)" % ", ".join(valu.get("files", [])[:3])
            valu["stale"] = bool(same_workload and per_sample_ms > 0 and abs(live_per_sample_ms / per_sample_ms - 1) > 0.02)
            if "issue_frac_2clk" in valu and "lane_occupancy" in valu:
                valu["lane_slot_frac"] = valu["issue_frac_2clk"] * valu["lane_occupancy"]   # useful lane-slots / (2-clock issue x 64 lanes): the kernel's own roofline fraction
            if same_workload and valu.get("hbm_bytes_per_launch") and valu.get("samples_per_launch"):
                traffic = valu["hbm_bytes_per_launch"] * (samples_per_step / world) / valu["samples_per_launch"]
        p0 = frames[0].params
        # the design-independent roofline: useful lane-instructions of the launch set over the measured kernel time, against the chip's lane-slots
        props = torch.cuda.get_device_properties(dev)
        peak_tlaneops = props.multi_processor_count * 128 * MAX_CLOCK_HZ / 1e12   # 256 CUs x 4 SIMDs x 32 lanes per clock x 2.4 GHz = 78.6 (half the guide's 157.3 TFLOP/s, which counts an FMA as two)
        wc = work_counters()
        model_ops, model_frames = 0.0, []
        for fr in frames:
            if fr.label == "veach_normal_aov":   # a first-hit pass of one sample per pixel: one traversal and a vertex
                u, terms = (2 + 11 * 20 + LANE_OP["traversal_setup"] + LANE_OP["vertex"]), None
            else:
                u, terms = useful_lane_ops(fr.label, wc) if both_mis else (None, None)
            if u is None:
                model_ops = None
                break
            model_ops += u * fr.samples / world
            model_frames.append({"frame": fr.label, "useful_lane_instr_per_sample": u, "terms": terms, "counters": wc.get(fr.label)})
        model_achieved = model_ops / (kernel_total_ms * 1e-3) / 1e12 if model_ops is not None else None
        # profiles/valu.json is only as good as the code it profiled: it carries kyhip_kernel_source_hash() of the library of its rocprofv3 session
        lib_sha = "%016x" % lib.kyhip_kernel_source_hash()
        if valu:
            valu["kernel_source_hash_profiled"] = valu.get("kernel_source_hash")
            valu["kernel_source_hash_loaded"] = lib_sha
            if valu.get("kernel_source_hash") != lib_sha:
                valu["stale"] = True
                valu["stale_reason"] = "the loaded library's kernels are not compiled from the sources profiles/valu.json was measured on (kyhip_kernel_source_hash): utilisation figures withheld"
                for k in ("lane_slot_frac", "issue_frac_2clk", "issue_frac_ubench", "issue_frac_mix_model", "lane_occupancy", "wave_instr_per_sample", "ns_per_valu_per_simd"):
                    valu[k] = None
                traffic = None
        # What ran: executed active lane-instructions per camera sample from the kernel's own counters (SQ_INSTS_VALU x 64 x lane occupancy), available when
        # profiles/valu.json describes the loaded kernels on this very workload.  `lane_slot_frac` prices them against the same peak over the LIVE kernel time; the
        # model's useful work must not exceed them (it is a floor: VERDICT round 4) -- if it ever does, the line says so and `frac` falls back to what ran.
        executed_per_sample = lane_slot_frac = None
        exceeds = None
        if valu and not valu.get("stale") and key == args.workload and valu.get("wave_instr_per_sample") and valu.get("lane_occupancy"):
            executed_per_sample = valu["wave_instr_per_sample"] * 64.0 * valu["lane_occupancy"]
            lane_slot_frac = executed_per_sample * (samples_per_step / world) / (kernel_total_ms * 1e-3) / 1e12 / peak_tlaneops
            if model_ops is not None:
                exceeds = bool(model_ops / (samples_per_step / world) > executed_per_sample)
        frac = (model_achieved / peak_tlaneops) if model_achieved is not None else None
        if exceeds:
            frac, model_achieved = lane_slot_frac, lane_slot_frac * peak_tlaneops
        line = {
            "schema": 6,   # round of the line's layout: since 5 `value` is the pipelined rate (`single_frame.value` continues rounds 1-4's series) and `roofline.frac` the floor model's
            "metric": "Msamples/s (paths*spp)", "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "launch_mode": "pipelined (consecutive frames on two streams)" if pipeline else "single frame (one stream, nothing overlapped)",
            "kernels": "the library's table (kyhip_set_jit(0) for everything timed here; this process started in mode %s)" % JIT_DEFAULT_MODE,
            "pipeline_priming_steps": R["priming"],   # untimed steps beyond `warmup` that brought the second stream's buffers into being (0 when warmup >= 2)
            "single_frame": ({"value": samples_per_step * args.steps / R1["elapsed"] / 1e6, "unit": "Msamples/s", "ms_per_step": R1["elapsed"] / args.steps * 1e3,
                              "steps": args.steps, "note": "the same steps with every frame on one stream: no overlap of a frame's start with the previous frame's tail"}
                             if R1 else ({"value": value, "unit": "Msamples/s", "ms_per_step": ms_per_step, "steps": args.steps} if not pipeline else None)),
            "ranks": ranks, "communicator": comm,
            "config": {"workload": name, "frames": [fr.label for fr in frames], "width": p0.width, "height": p0.height, "spp": p0.samples_per_pixel,
                       "max_path_depth": p0.max_path_depth, "direct_sample": {0: "idle", 4: "bsdf", 8: "light", 16: "bsdf_mis", 32: "light_mis", 48: "both_mis"}[p0.direct_sample],
                       "integrator": {6: "direct_lighting", 8: "simple_path_tracing_recursion", 9: "path_tracing_recursion", 10: "path_tracing_recursion_defered", 11: "path_tracing_iteration"}[args.integrator],
                       "seed": p0.seed, "tile": [p0.tile_w, p0.tile_h],
                       "parallelism": "image tiles interleaved over %d GPU(s), one film-tile gather per frame; launches %s" % (world, "pipelined on two streams" if pipeline else "on one stream")},
            # The kernel keeps all path state in registers and LDS and is bound by VALU issue, so the roofline that bounds it is lane-slots:
            # `achieved` = USEFUL lane-instructions (the reference's event counts x the least work that decides each, `valu_model`: a floor under what any
            # kernel must execute) over the kernel time measured live with HIP events; `peak` = CUs x 4 SIMDs x 32 lanes per clock x the maximum engine
            # clock; `lane_slot_frac` = the active lane-instructions the kernel really executed (its counters, profiles/valu.json) over the same peak and
            # time: frac <= lane_slot_frac <= 1, the gap between the two is instructions that do no path arithmetic.  `contract` keeps SURVEY 8(d)'s figure -- ALGORITHMIC bytes of an HBM
            # ray-pool tracer over the same kernel time against 8 TB/s -- which stopped bounding anything at 14.7 Gsamples/s because those bytes
            # never move here; `traffic` is what the memory side really saw (profiled build only); `valu` are the kernel's own utilisation counters.
            "roofline": {"bound": "valu", "achieved": model_achieved, "peak": peak_tlaneops, "unit": "Tlaneop/s (useful fp32 / int32 VALU lane-instructions per second)",
                         "frac": frac,
                         "lane_slot_frac": lane_slot_frac, "executed_lane_instr_per_sample": executed_per_sample,
                         "useful_lane_instr_per_sample": (model_ops / (samples_per_step / world)) if model_ops is not None else None,
                         "valu_model_exceeds_executed": exceeds,
                         #  why the check above is None when it is: the counter set of profiles/valu.json describes other kernel sources or another workload
                         "valu_counters_status": ("current" if exceeds is not None else
                                                  ("none for this workload" if not valu else (valu.get("stale_reason") or ("stale: kernel time moved" if valu.get("stale") else "no executed figure")))),
                         "traffic": traffic, "traffic_source": (valu or {}).get("source") if traffic is not None else None,
                         "kernel": "render_kernel", "kernel_ms": kernel_total_ms, "kernel_ms_per_frame": frame_kernel_ms,
                         "samples_per_launch_set": samples_per_step // world,
                         "valu_model": {"frames": model_frames, "lane_op_costs": LANE_OP, "counters_source": "tests/golden/work_counters.json (CPU oracle; SURVEY section 6 for the reference's own)",
                                        "peak_definition": "%d CUs x 4 SIMDs x 32 lanes/clk x %.2f GHz (maximum engine clock)" % (props.multi_processor_count, MAX_CLOCK_HZ / 1e9)} if model_ops is not None else None,
                         "contract": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                                      "algorithmic_bytes_per_sample": launch_bytes * world / samples_per_step,
                                      "note": "SURVEY 8(d): 128 B per path iteration + 12 B of film per sample; a throughput-equivalent, these bytes never move"},
                         "valu": valu},
            "film_mean": R["film_mean"],
        }
        default_line = world == 1 and args.workload == "cornell" and both_mis and not (args.width or args.height or args.spp or args.depth)
        if default_line and not args.no_extra:
            line["extra_workloads"] = extra_workloads(args, run_workload, peak_tlaneops, lib, local_rank, kernel_total_ms, parity=not args.no_cpu_baseline)
            line["projected_scaling"] = projected_scaling(frames[0], dev, local_rank, lib, frame_kernel_ms[0], ms_per_step)
        if default_line and not args.no_extra:
            line["boundary"] = boundary_rates(frames[0], dev, local_rank)
        if world == 1 and not args.no_cpu_baseline:
            def gpu_render(scene, sample_params):
                return api.render(scene, sample_params, device=local_rank)
            cb, extra = cpu_baseline(frames, gpu_render, args.cpu_seconds, args.workload)
            line["cpu_baseline"] = cb
            line.update(extra)
            if cb["omp_threads"] <= cb["cpus_granted"]:
                line["speedup_vs_cpu_baseline"] = value / cb["value"]
        # parity gate of the line: every RMSE at full spp the line carries must be under the north star's 1e-3 -- without the pixels whose
        # differing samples are all explained, and under 2e-3 with them -- and no pixel may be off by more than 5e-3 without its samples explaining it; otherwise the
        # line is printed and the process exits non-zero
        gates = [("headline", line.get("rmse_full_spp"))] + [(k, v.get("rmse_full_spp")) for k, v in (line.get("extra_workloads") or {}).items() if isinstance(v, dict)]
        failed = [k for k, g in gates if g and (not (g.get("value_without_explained", g["value"]) < g["target"]) or not (g["value"] < 2 * g["target"]))]
        # (a pixel the replay could not classify -- or did not get to -- is reported for review and does not fail the line: its difference is INSIDE the figures above)
        review = [k for k, g in gates if g and (g.get("unexplained", 0) > 0 or g.get("not_examined", 0) > 0)]
        line["parity_gate"] = {"checked": [k for k, g in gates if g], "failed": failed, "review": review,
                               "rule": "failed (exit code 3): rmse_full_spp.value_without_explained >= 1e-3 or .value >= 2e-3; review: unexplained or unexamined pixels"}
        print(json.dumps(line), flush=True)
        if failed:
            sys.stderr.write("bench.py: parity gate failed for %s\n" % ", ".join(failed))
            if world > 1:
                tdist.destroy_process_group()
            sys.exit(3)
    if world > 1:
        tdist.destroy_process_group()


def extra_workloads(args, run_workload, peak_tlaneops, lib, local_rank, kernel_ms_specialised, parity=True):
    """configs[2], [3] and [4], one timed step each (outside the headline's timed region; veach and batch after a warm-up step, the
    25-second stress frame without one), so that the driver's record of the default run carries them: value, ms per step, the live kernel
    duration of every frame, the contract fraction."""
    out = {}
    O = threads = None
    if parity:
        from oracle import kyoracle as O   # the checker of rmse_full_spp below; never inside a timed region
        threads = max(1, min(O.max_threads(), cpus_granted()))
    for wl in ("veach", "batch", "stress", "single"):
        wargs = argparse.Namespace(**vars(args))
        wargs.workload = wl
        wargs.width = wargs.height = wargs.spp = wargs.depth = 0
        warm = 0 if wl == "stress" else 1
        r = run_workload(wargs, 1, warm, record_in_timed=(wl == "stress"))
        samples = sum(fr.samples for fr in r["frames"])
        achieved, _ = contract_frac(r["frames"], r["kernel_ms"])
        wc = work_counters()
        ops = 0.0
        for fr in r["frames"]:
            u = (2 + 11 * 20 + LANE_OP["traversal_setup"] + LANE_OP["vertex"]) if fr.label == "veach_normal_aov" else useful_lane_ops(fr.label, wc)[0]
            ops = None if (ops is None or u is None) else ops + u * fr.samples
        # what ran (the workload's own counter set of profiles/valu.json, when it describes the loaded kernels) beside the model's floor; a model above it is reported, and capped
        ex = executed_lane_ops(wl, "%016x" % lib.kyhip_kernel_source_hash())
        k_s = sum(r["kernel_ms"]) * 1e-3
        lane_slot = (ex * samples / k_s / 1e12 / peak_tlaneops) if ex is not None else None
        model_frac = (ops / k_s / 1e12 / peak_tlaneops) if ops is not None else None
        exceeds = bool(model_frac > lane_slot) if (model_frac is not None and lane_slot is not None) else None
        out[wl] = {"workload": r["name"], "value": samples / r["elapsed"] / 1e6, "unit": "Msamples/s", "steps": 1, "warmup": warm, "ms_per_step": r["elapsed"] * 1e3,
                   "frames": [fr.label for fr in r["frames"]], "kernel_ms_per_frame": r["kernel_ms"], "kernel_ms": sum(r["kernel_ms"]),
                   "roofline_frac": lane_slot if exceeds else model_frac, "lane_slot_frac": lane_slot, "valu_model_exceeds_executed": exceeds,
                   "contract_frac": achieved / HBM_PEAK_GBS, "film_mean": r["film_mean"]}
        if parity:
            # parity in the driver-run line: GPU against the CPU oracle at the config's FULL spp on an interleaved tile sample of at least
            # 4096 pixels -- fewer (never under 1024) where the oracle would need more than about 40 s of this host's CPUs for them (the 16 384-spp frame on 2 CPUs)
            spp_sum = sum(fr.params.samples_per_pixel for fr in r["frames"] if fr.params.samples_per_pixel > 1)
            n_path = max(1, sum(1 for fr in r["frames"] if fr.params.samples_per_pixel > 1))
            cpu_rate = (0.45e6 if wl == "veach" else 0.9e6) * threads          # the oracle's samples per second per thread, roughly (this box measures its own below)
            px = int(min(4096, max(1024, 40.0 * cpu_rate / (spp_sum / n_path))))
            out[wl]["rmse_full_spp"] = parity_full_spp(r["frames"], lambda scene, sp: api.render(scene, sp, device=local_rank), O, threads, min_pixels=px)
    # configs[1] without the scene-fact instantiations (kyhip_set_specialisation(0): what a scene outside the table of facts gets -- the
    # both_mis kernel that assumes nothing about lights or materials), next to its specialised twin of the headline
    prev = lib.kyhip_set_specialisation(0)
    try:
        wargs = argparse.Namespace(**vars(args))
        wargs.width = wargs.height = wargs.spp = wargs.depth = 0
        r = run_workload(wargs, 1, 1)
        samples = sum(fr.samples for fr in r["frames"])
        out["cornell_unspecialised"] = {"workload": r["name"] + " [kyhip_set_specialisation(0)]", "value": samples / r["elapsed"] / 1e6, "unit": "Msamples/s", "steps": 1, "warmup": 1,
                                        "ms_per_step": r["elapsed"] * 1e3, "kernel": lib.kyhip_last_kernel(local_rank).decode(), "kernel_ms": sum(r["kernel_ms"]),
                                        "ratio_to_specialised_kernel": kernel_ms_specialised / sum(r["kernel_ms"]), "film_mean": r["film_mean"]}
    finally:
        lib.kyhip_set_specialisation(prev)
    # Run-time instantiations (kyhip_set_jit, off by default): the point-light Cornell box on the kernel compiled for ALL of its scene's facts (the table's
    # row knows "one delta light"), and a scene the table has no row for -- the Cornell box lit by its lamp AND the point light, both_mis: the fact-free
    # kernel in the table, this scene's facts when instantiated (shadow rays inline either way: ky_pack.cpp, shadow_queue_wanted) -- each against the table's kernel.  (The first launch of an instantiation compiles it: a warm-up step.)
    import torch
    dev = torch.device("cuda", local_rank)

    def kernel_ms_of(scene, p, reps=3):
        film = torch.zeros((p.height, p.width, 3), dtype=torch.float32, device=dev)
        best = float("inf")
        for _ in range(reps + 1):
            kydist.render_distributed(scene, p, 0, 1, local_rank, film=film)
            torch.cuda.synchronize(dev)
            best = min(best, float(lib.kyhip_kernel_ms(local_rank)))
        return best, lib.kyhip_last_kernel(local_rank).decode().split(" (")[0]

    W, H = 1024, 768
    two = api.cornell_box_scene(A.CB_BOTH_SMALL_SPHERES | A.CB_LIGHT_AREA | A.CB_LIGHT_POINT, W, H)
    cases = {"cornell_point_light": (api.cornell_box_scene(A.CB_BOTH_SMALL_SPHERES | A.CB_LIGHT_POINT, W, H), api.make_params(W, H, 256)),   # the table knows "one delta light"; the scene has more facts
             "cornell_lamp_and_point_light": (two, api.make_params(W, H, 256))}
    jit = {}
    prev = lib.kyhip_set_jit(-1)
    jit["default_mode"] = JIT_DEFAULT_MODE   # what this process started in (round 6: 2 = asynchronous, for a single-process job with a compiler at hand and no profiler attached)
    try:
        for label, (scene, p) in cases.items():
            lib.kyhip_set_jit(0)
            t_ms, t_kernel = kernel_ms_of(scene, p)
            # mode 2, as a caller outside the table meets it: frames render on the table's kernel while a background thread compiles, and switch at a frame boundary
            lib.kyhip_set_jit(2)
            t0, frames_on_table, switched = time.perf_counter(), 0, False
            film = torch.zeros((p.height, p.width, 3), dtype=torch.float32, device=dev)
            while time.perf_counter() - t0 < 60.0:
                kydist.render_distributed(scene, p, 0, 1, local_rank, film=film)
                torch.cuda.synchronize(dev)
                if "run-time instantiation" in lib.kyhip_last_kernel(local_rank).decode():
                    switched = True
                    break
                frames_on_table += 1
                time.sleep(0.05)
            wait_s = time.perf_counter() - t0
            o_ms, o_kernel = kernel_ms_of(scene, p) if switched else (float("nan"), lib.kyhip_last_kernel(local_rank).decode())
            samples = p.width * p.height * p.samples_per_pixel
            jit[label] = {"spp": p.samples_per_pixel, "table": {"kernel": t_kernel, "kernel_ms": t_ms, "value": samples / t_ms / 1e3},
                          "own": {"kernel": o_kernel, "kernel_ms": o_ms, "value": samples / o_ms / 1e3}, "unit": "Msamples/s (kernel time)", "own_over_table": t_ms / o_ms,
                          "mode_2": {"frames_on_the_table_kernel": frames_on_table, "seconds_until_own_kernel": wait_s if switched else None, "switched": switched}}
        jit["status"] = lib.kyhip_jit_status().decode()
    finally:
        lib.kyhip_set_jit(prev)
    out["run_time_instantiations"] = jit
    return out


def boundary_rates(fr, dev, local_rank):
    """What the seam itself delivers: kyhip_render (the C-ABI call behind integrator_t::render(&scene, sampler, &film): HOST film in, host
    film out, blocking -- the call the reference times, ky.cpp:4695-4698) on configs[1] at its own spp and at 64 spp, next to the
    device-resident path (kyhip_render_tiles_device + one add kernel, film in HBM) timed the same way: wall clock around each of `reps` blocking calls, the median reported."""
    out = []
    film_host = np.zeros((fr.params.height, fr.params.width, 3), np.float32)
    film_dev = torch.zeros((fr.params.height, fr.params.width, 3), dtype=torch.float32, device=dev)
    for spp, reps in ((fr.params.samples_per_pixel, 3), (64, 11)):
        p = A.RenderParams.from_buffer_copy(fr.params)
        p.samples_per_pixel = spp
        samples = p.width * p.height * spp
        def per_call(call):
            """(median, mean) ms of `reps` blocking calls timed one by one, after one warm-up call.  The median is the figure: on a box that grants two CPUs another
            tenant's burst lands in one call of ten now and then (round 6: one run's ten 64-spp calls averaged 4.7 ms where four other runs of the same build measured 3.3), and it says what a call costs;
            the mean stays in the line next to it."""
            call()
            ms = []
            for _ in range(reps):
                t0 = time.perf_counter()
                call()
                ms.append((time.perf_counter() - t0) * 1e3)
            return float(np.median(ms)), float(np.mean(ms))

        def device_resident():
            kydist.render_distributed(fr.scene, p, 0, 1, local_rank, film=film_dev)
            torch.cuda.synchronize(dev)
        host_ms, host_mean = per_call(lambda: api.render(fr.scene, p, film=film_host, device=local_rank))   # (the warm-up call also sizes the seam's cached buffers)
        dev_ms, dev_mean = per_call(device_resident)
        # the same call into a PINNED film (kyhip_film_alloc: what ky.hpp's film_t owns): the GPU adds to it in place -- no staging copy, no host pass
        pinned = api.PinnedFilm(fr.params.height, fr.params.width)
        pinned_ms, pinned_mean = per_call(lambda: api.render(fr.scene, p, film=pinned.array, device=local_rank))
        del pinned
        out.append({"spp": spp, "calls": reps, "statistic": "median of the calls, each timed by itself (mean alongside)",
                    "ms_per_call": host_ms, "ms_per_call_mean": host_mean, "value": samples / host_ms / 1e3, "unit": "Msamples/s",
                    "device_resident_ms_per_call": dev_ms, "device_resident_ms_per_call_mean": dev_mean, "device_resident_value": samples / dev_ms / 1e3,
                    "ratio_to_device_resident": dev_ms / host_ms,
                    "pinned_film_ms_per_call": pinned_ms, "pinned_film_ms_per_call_mean": pinned_mean, "pinned_film_value": samples / pinned_ms / 1e3,
                    "pinned_film_ratio_to_device_resident": dev_ms / pinned_ms})
    lib = A.load_kyhip()
    return {"entry": "kyhip_render (host film in / out, blocking; include/kyhip.h)", "film_bytes": int(film_host.nbytes), "rates": out,
            "host_threads_adding": int(lib.kyhip_seam_threads()), "cpus_granted": cpus_granted(), "status": lib.kyhip_multi_status(local_rank).decode()}


def projected_scaling(fr, dev, local_rank, lib, kernel_ms_n1, ms_per_step_n1):
    """Strong-scaling projection from ONE GPU: every rank of an N-GPU run renders a 1/N shard of the frame with no communication, so the
    slowest shard's kernel sets the frame time.  Renders all N shards for N = 2, 4, 8 here, one after the other:
      kernel_efficiency[N] = kernel_ms(N = 1) / (N x max_r kernel_ms(shard r of N)).
    The gather of W*H*12/N bytes per rank and the add kernel come on top (microseconds; never measured on hardware).
    pipelined: a persistent kernel pays start-up and tail once per launch; with launches on two alternating streams the next frame fills
    them.  shard_ms_pipelined = time per frame of 16 back-to-back launches of the slowest N = 8 shard."""
    out = {"n": [], "kernel_efficiency": [], "slowest_shard_kernel_ms": [], "kernel_ms_n1": kernel_ms_n1}
    slowest8 = 0
    for n in (2, 4, 8):
        buf = torch.zeros((kydist.shard_tile_count(fr.params, 0, n), fr.params.tile_h, fr.params.tile_w, 3), dtype=torch.float32, device=dev)
        ms = []
        for r in range(n):
            best = float("inf")
            for _ in range(2):   # the faster of two launches per shard: a shard's kernel time, not the box's jitter (the slowest SHARD still sets the pace)
                kydist.render_shard(fr.scene, fr.params, r, n, local_rank, out=buf)
                torch.cuda.synchronize(dev)
                best = min(best, float(lib.kyhip_kernel_ms(local_rank)))
            ms.append(best)
        out["n"].append(n)
        out["slowest_shard_kernel_ms"].append(max(ms))
        out["kernel_efficiency"].append(kernel_ms_n1 / (n * max(ms)))
        if n == 8:
            slowest8 = int(np.argmax(ms))
    streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
    bufs = [torch.zeros_like(buf), torch.zeros_like(buf)]
    reps = 16

    def burst(k):
        for i in range(k):
            with torch.cuda.stream(streams[i & 1]):
                kydist.render_shard(fr.scene, fr.params, slowest8, 8, local_rank, out=bufs[i & 1])
    burst(2)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    burst(reps)
    torch.cuda.synchronize(dev)
    shard_ms = (time.perf_counter() - t0) / reps * 1e3
    out["pipelined"] = {"n": 8, "shard_ms_pipelined": shard_ms, "frame_ms_n1_pipelined": ms_per_step_n1, "efficiency": ms_per_step_n1 / (8 * shard_ms)}
    # configs[4] is the frame the north star actually spreads over 8 GPUs (4096 x 4096, depth 16): its shards hold 21 times the pixels of
    # configs[1]'s, so a launch's start-up and drain weigh that much less.  Shown at 1/64 of its 16 384 spp (the chunk schedule's bulk is
    # the same from 256 spp up; the full frame takes 21 s): the whole frame, then every 1/8 shard, on this one GPU.
    sw, sh_, sspp = 4096, 4096, 256
    sscene = api.cornell_box_scene(A.CB_DEFAULT_SCENE, sw, sh_)
    sp = api.make_params(sw, sh_, sspp, max_path_depth=16)
    sbuf = torch.zeros((kydist.shard_tile_count(sp, 0, 1), sp.tile_h, sp.tile_w, 3), dtype=torch.float32, device=dev)
    kydist.render_shard(sscene, sp, 0, 1, local_rank, out=sbuf)
    torch.cuda.synchronize(dev)
    full_ms = float(lib.kyhip_kernel_ms(local_rank))
    ms = []
    for r in range(8):
        kydist.render_shard(sscene, sp, r, 8, local_rank, out=sbuf)
        torch.cuda.synchronize(dev)
        ms.append(float(lib.kyhip_kernel_ms(local_rank)))
    out["stress"] = {"frame": "configs[4] geometry (Cornell 4096x4096, depth 16) at %d spp" % sspp, "n": 8, "kernel_ms_n1": full_ms, "slowest_shard_kernel_ms": max(ms),
                     "shard_kernel_ms": ms, "kernel_efficiency": full_ms / (8 * max(ms))}
    return out


if __name__ == "__main__":
    main()
