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
            One "step" renders the workload's frame(s) once.
  N > 1     every frame's tiles are interleaved over the N ranks (one process per GPU), each rank renders its tiles with no
            communication, then ONE gather of film tiles to rank 0 (RCCL over xGMI) and one add kernel there.  Total work is
            fixed, so scaling is "strong".

Launch: `python bench.py` (N=1) or
        `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
             bench.py --gpus N --steps K --warmup W`.
Rank 0 prints ONE JSON line.  The CPU oracle is used here ONLY for the reported `cpu_baseline` (and the RMSE next to it); it is
never inside the timed GPU region.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from ky_amd import _abi as A  # noqa: E402
from ky_amd import api, dist as kydist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s


def bytes_per_sample(iterations):
    """algorithmic bytes per camera sample of an HBM ray-pool tracer = 128 B x mean path iterations + 12 B film."""
    return 128.0 * iterations + 12.0


# mean path iterations per camera sample (scene->intersect calls made by Li): SURVEY.md section 6 for configs[1] / [2]
# (4.168 / 2.711, measured on the reference); the other frames with the oracle's counters (oracle/ky_oracle.cpp, 128x128x64):
# Cornell at depth 16: 4.266 (roulette ends paths long before the cap); point / direction / environment lights: 4.215;
# Veach at 1:1 aspect: 2.803; a first-hit AOV pass: 1.
ITER = {"cornell": 4.168, "veach": 2.711, "cornell_d16": 4.266, "cornell_other_lights": 4.215, "veach_square": 2.803, "aov": 1.0}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cornell", choices=["cornell", "veach", "batch", "stress"])
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--spp", type=int, default=0)
    ap.add_argument("--depth", type=int, default=0)
    ap.add_argument("--direct-sample", type=int, default=A.DIRECT_BOTH_MIS, choices=[0, 4, 8, 16, 32, 48],
                    help="direct_sample_enum_t of the path integrator frames (profiling the run-time-dispatched kernel; the metric's configs use 48)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the cpu_baseline sample")
    return ap.parse_args()


class Frame:
    def __init__(self, label, scene, params, iterations, origin=(0, 0)):
        self.label, self.scene, self.params, self.origin = label, scene, params, origin
        self.bytes_per_sample = bytes_per_sample(iterations)
        self.samples = params.width * params.height * params.samples_per_pixel


def workload(args):
    """-> (frames, film (height, width), name)"""
    if args.workload == "cornell":
        W, H, spp, depth = args.width or 1024, args.height or 768, args.spp or 1024, args.depth or 5
        frames = [Frame("cornell", api.cornell_box_scene(A.CB_DEFAULT_SCENE, W, H), api.make_params(W, H, spp, max_path_depth=depth, direct_sample=args.direct_sample), ITER["cornell"])]
        name = "BASELINE configs[1]: ky Cornell box (both_small_spheres|light_area) %dx%d, %d spp, path_tracing_iteration d%d both_mis" % (W, H, spp, depth)
        return frames, (H, W), name
    if args.workload == "veach":
        W, H, spp, depth = args.width or 1280, args.height or 720, args.spp or 4096, args.depth or 5
        frames = [Frame("veach", api.mis_scene(W, H), api.make_params(W, H, spp, max_path_depth=depth, direct_sample=args.direct_sample), ITER["veach"])]
        name = "BASELINE configs[2]: ky Veach MIS scene (create_mis_scene) %dx%d, %d spp, path_tracing_iteration d%d both_mis" % (W, H, spp, depth)
        return frames, (H, W), name
    if args.workload == "stress":
        W, H, spp, depth = args.width or 4096, args.height or 4096, args.spp or 16384, args.depth or 16
        frames = [Frame("cornell_d16", api.cornell_box_scene(A.CB_DEFAULT_SCENE, W, H), api.make_params(W, H, spp, max_path_depth=depth, direct_sample=args.direct_sample), ITER["cornell_d16"])]
        name = "BASELINE configs[4]: stress, ky Cornell box %dx%d, %d spp, path_tracing_iteration d%d both_mis" % (W, H, spp, depth)
        return frames, (H, W), name
    # batch: render_multiple_scene (ky.cpp:4819-4876) at production size, one film_grid_t(2, 3, res, res)
    res, spp, depth = args.width or 1024, args.spp or 2048, args.depth or 5
    frames = []
    lights = (("point", A.CB_LIGHT_POINT), ("direction", A.CB_LIGHT_DIRECTION), ("area", A.CB_LIGHT_AREA), ("environment", A.CB_LIGHT_ENVIRONMENT))
    for cell, (lname, flag) in enumerate(lights):
        it = ITER["cornell"] if lname == "area" else ITER["cornell_other_lights"]
        frames.append(Frame("cornell_" + lname, api.cornell_box_scene(A.CB_BOTH_SMALL_SPHERES | flag, res, res), api.make_params(res, res, spp, max_path_depth=depth, direct_sample=args.direct_sample),
                            it, ((cell % 3) * res, (cell // 3) * res)))
    veach = api.mis_scene(res, res)
    frames.append(Frame("veach", veach, api.make_params(res, res, spp, max_path_depth=depth, direct_sample=args.direct_sample), ITER["veach_square"], (res, res)))
    frames.append(Frame("veach_normal_aov", veach, api.make_params(res, res, 1, integrator=A.INTEGRATOR_NORMAL, sampler=A.SAMPLER_DEBUG), ITER["aov"], (2 * res, res)))
    name = ("BASELINE configs[3]: render_multiple_scene batch, 4 Cornell light variants + Veach (path_tracing_iteration d%d both_mis) + first-hit AOV, "
            "each %dx%d at %d spp, film_grid 2x3" % (depth, res, res, spp))
    return frames, (2 * res, 3 * res), name


def cpu_baseline(frames, gpu_render, target_seconds):
    """Time the CPU oracle (a port of the reference's algorithm, all host threads) on a BOUNDED sample of the same workload:
    of every frame an interleaved subset of its tiles (about 131k pixels spread over the whole picture; the oracle honours
    tile_first / tile_step) at reduced spp -- the rate depends on neither.  Returns the baseline object and the RMSE of the
    GPU's render of the very same sample against it."""
    from oracle import kyoracle as O
    threads = O.max_threads()
    # what the host really grants this process: the pool's boxes differ (some confine a job to a few cores' worth of CPU time
    # whatever the thread count) -- reported next to the thread count so that the baseline can be read
    granted = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            granted = min(granted, max(1, int(round(int(quota) / int(period)))))
    except Exception:
        pass

    def sample_params(fr, spp):
        p = A.RenderParams.from_buffer_copy(fr.params)
        p.tile_first = 0
        p.tile_step = max(1, kydist.tiles_total(fr.params) * fr.params.tile_w * fr.params.tile_h // 131072)
        p.samples_per_pixel = max(1, min(fr.params.samples_per_pixel, spp))
        return p

    def run(spp_scale):
        films, n, t = [], 0, 0.0
        for fr in frames:
            p = sample_params(fr, int(round(fr.params.samples_per_pixel * spp_scale)))
            t0 = time.perf_counter()
            film = O.render(fr.scene, p)
            t += time.perf_counter() - t0
            owned = kydist.shard_tile_count(p, 0, p.tile_step) * p.tile_w * p.tile_h   # includes the few pixels of edge tiles outside the film
            n += min(owned, p.width * p.height) * p.samples_per_pixel
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
    # the GPU renders the very same sample; the reference's own arithmetic yields inf * 0 = NaN for a few exactly-grazing mirror
    # hits (DESIGN.md "Non-finite samples"); such pixels are excluded from the RMSE and counted
    se, cnt, bad = 0.0, 0, 0
    for fr, (p, cpu_film) in zip(frames, films):
        gpu_film = gpu_render(fr.scene, p)
        lit = (cpu_film != 0).any(axis=2) | (gpu_film != 0).any(axis=2)
        fin = np.isfinite(cpu_film).all(axis=2) & np.isfinite(gpu_film).all(axis=2)
        d = gpu_film[fin & lit].astype(np.float64) - cpu_film[fin & lit].astype(np.float64)
        se += float((d * d).sum()); cnt += d.size; bad += int((~fin).sum())
    spps = sorted({p.samples_per_pixel for p, _ in films})
    return {
        "value": n / dt / 1e6, "unit": "Msamples/s", "cores": min(threads, granted), "kind": "port", "omp_threads": threads, "cpus_granted": granted,
        "sample": "%d frame(s) of the workload, every %d-th tile (interleaved over the picture), %s spp: %.3g samples, %.1f s of CPU work, OpenMP %d threads"
                  % (len(frames), films[0][0].tile_step, "/".join(map(str, spps)), n, dt, threads),
    }, {"rmse_gpu_vs_cpu": (se / max(cnt, 1)) ** 0.5, "rmse_spp": spps[-1], "rmse_excluded_nonfinite_pixels": bad}


def load_json(name):
    try:
        with open(os.path.join(ROOT, "profiles", name)) as fh:
            return json.load(fh)
    except Exception:
        return None


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
    # KY_BENCH_ONE_GPU=1 (testing only): every rank uses cuda:0 and the gather runs over gloo, so that the N > 1 code
    # path can be exercised on a single-GPU box; the numbers of such a run are meaningless.
    one_gpu_test = os.environ.get("KY_BENCH_ONE_GPU") == "1"
    if one_gpu_test:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as tdist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_gpu_test:
            tdist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            tdist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    frames, (FH, FW_), name = workload(args)
    film = torch.zeros((FH, FW_, 3), dtype=torch.float32, device=dev) if rank == 0 else None

    def barrier():
        if world > 1:
            tdist.barrier()
        torch.cuda.synchronize(dev)

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
            kydist.render_distributed(fr.scene, fr.params, rank, world, local_rank, film=target(fr))
            if record:
                torch.cuda.current_stream(dev).synchronize()
                kernel_ms[i].append(float(lib.kyhip_kernel_ms(local_rank)))

    for _ in range(args.warmup):
        step(False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(False)
    barrier()
    elapsed = time.perf_counter() - t0
    # kernel durations (HIP events recorded by the library on the launch stream around render_kernel only): a second, untimed
    # pass with a sync after every frame so that each event pair is read back
    for _ in range(1 if frames[0].samples > 2e10 else max(1, min(args.steps, 3))):
        step(True)
    barrier()

    t = torch.tensor([elapsed] + [sum(k) / len(k) for k in kernel_ms], dtype=torch.float64, device=dev)
    if world > 1:
        tdist.all_reduce(t, op=tdist.ReduceOp.MAX)
    elapsed = float(t[0].item())
    frame_kernel_ms = [float(v) for v in t[1:].tolist()]

    if rank == 0:
        samples_per_step = sum(fr.samples for fr in frames)
        ms_per_step = elapsed / args.steps * 1e3
        value = samples_per_step * args.steps / elapsed / 1e6
        # roofline of the dominant kernel (render_kernel): one launch per frame; the figure is over the step's launches
        launch_bytes = sum(fr.bytes_per_sample * fr.samples / world for fr in frames)
        kernel_total_ms = sum(frame_kernel_ms)
        achieved = launch_bytes / (kernel_total_ms * 1e-3) / 1e9
        tj = load_json("hbm_traffic.json")
        traffic = None
        if tj and tj.get("workload") == args.workload and tj.get("samples_per_launch") and frames[0].params.direct_sample == A.DIRECT_BOTH_MIS:
            traffic = tj["hbm_bytes_per_launch"] * (samples_per_step / world) / tj["samples_per_launch"]
        vj = load_json("valu.json")
        both_mis = frames[0].params.direct_sample == A.DIRECT_BOTH_MIS
        valu = vj.get(args.workload) if (vj and both_mis) else None
        if vj and not both_mis and args.workload == "cornell" and frames[0].params.direct_sample == A.DIRECT_LIGHT_MIS:
            valu = vj.get("light_mis")
        if valu is None and vj and both_mis and args.workload in ("stress", "batch"):
            valu = dict(vj.get("cornell") or {}, note="counters of the cornell workload (same kernel, same scene family)")
        p0 = frames[0].params
        line = {
            "metric": "Msamples/s (paths*spp)", "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": name, "frames": [fr.label for fr in frames], "width": p0.width, "height": p0.height, "spp": p0.samples_per_pixel,
                       "max_path_depth": p0.max_path_depth, "direct_sample": {0: "idle", 4: "bsdf", 8: "light", 16: "bsdf_mis", 32: "light_mis", 48: "both_mis"}[p0.direct_sample], "seed": p0.seed,
                       "tile": [p0.tile_w, p0.tile_h], "parallelism": "image tiles interleaved over %d GPU(s), one film-tile gather per frame" % world},
            # `achieved` / `peak` / `frac` follow the contract of SURVEY.md 8(d): ALGORITHMIC bytes of an HBM ray-pool tracer (128 B per
            # path iteration + 12 B of film per sample) over the measured kernel time, against the HBM peak.  The kernel built here
            # keeps all path state in registers and LDS, so those bytes never move: `traffic` is what the memory side really saw, and
            # `bound` names what really limits the kernel -- VALU issue, quantified in `valu` (profiles/valu.json, tools/make_valu_json.py).
            "roofline": {"bound": "valu", "contract_bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "kernel": "render_kernel", "kernel_ms": kernel_total_ms, "kernel_ms_per_frame": frame_kernel_ms,
                         "algorithmic_bytes_per_sample": launch_bytes * world / samples_per_step, "samples_per_launch_set": samples_per_step // world,
                         "valu": valu},
            "film_mean": float(film.mean().item()),
        }
        if world == 1 and not args.no_cpu_baseline:
            def gpu_render(scene, sample_params):
                return api.render(scene, sample_params, device=local_rank)
            cb, extra = cpu_baseline(frames, gpu_render, args.cpu_seconds)
            line["cpu_baseline"] = cb
            line.update(extra)
            line["speedup_vs_cpu_baseline"] = value / cb["value"]
        print(json.dumps(line), flush=True)
    if world > 1:
        tdist.destroy_process_group()


if __name__ == "__main__":
    main()
