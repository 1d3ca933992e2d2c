#!/usr/bin/env python3
"""bench.py -- headline metric of BASELINE.json on its named configuration.

  metric   Msamples/s (paths x spp): camera samples per wall second, whole job.
  workload configs[1]: ky Cornell box (both_small_spheres | light_area), 1024x768, 1024 spp,
           path_tracing_iteration depth 5, both_mis -- one "step" renders that whole frame once.
  N > 1    the film's tiles are interleaved over the N ranks (one process per GPU), each rank renders its
           tiles with no communication, then ONE gather of film tiles to rank 0 (RCCL over xGMI), which adds
           them into the film.  Total work is fixed, so scaling is "strong".

Launch: `python bench.py` (N=1) or
        `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
             bench.py --gpus N --steps K --warmup W`.
Rank 0 prints ONE JSON line.  The CPU oracle is used here ONLY for the reported `cpu_baseline` (and the RMSE
next to it); it is never inside the timed GPU region.
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

# algorithmic bytes per camera sample = 128 B x mean path iterations + 12 B film
BYTES_PER_SAMPLE = {"cornell": 128 * 4.168 + 12, "veach": 128 * 2.711 + 12}
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cornell", choices=["cornell", "veach"])
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--spp", type=int, default=0)
    ap.add_argument("--depth", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the cpu_baseline sample")
    return ap.parse_args()


def workload(args):
    if args.workload == "cornell":
        W, H, spp = args.width or 1024, args.height or 768, args.spp or 1024
        scene = api.cornell_box_scene(A.CB_DEFAULT_SCENE, W, H)
        name = "ky Cornell box (both_small_spheres|light_area) %dx%d, %d spp, path_tracing_iteration d%d both_mis" % (W, H, spp, args.depth)
    else:
        W, H, spp = args.width or 1280, args.height or 720, args.spp or 4096
        scene = api.mis_scene(W, H)
        name = "ky Veach MIS scene %dx%d, %d spp, path_tracing_iteration d%d both_mis" % (W, H, spp, args.depth)
    params = api.make_params(W, H, spp, max_path_depth=args.depth)
    return scene, params, name


def cpu_baseline(scene, params, gpu_film_fn, target_seconds):
    """Time the CPU oracle (a port of the reference's algorithm, all host threads) on a bounded sample of the same
    workload: the same frame at reduced spp (the rate does not depend on spp)."""
    from oracle import kyoracle as O
    threads = O.max_threads()
    probe = A.RenderParams.from_buffer_copy(params)
    probe.samples_per_pixel = 1
    O.render(scene, probe)                      # warm-up: thread pool, page faults
    probe.samples_per_pixel = 4
    t0 = time.perf_counter()
    O.render(scene, probe)
    t4 = time.perf_counter() - t0
    spp = int(max(4, min(params.samples_per_pixel, round(target_seconds / max(t4 / 4, 1e-3)))))
    sample = A.RenderParams.from_buffer_copy(params)
    sample.samples_per_pixel = spp
    t0 = time.perf_counter()
    cpu_film = O.render(scene, sample)
    dt = time.perf_counter() - t0
    n = params.width * params.height * spp
    gpu_film = gpu_film_fn(sample)
    # the reference's own arithmetic yields inf * 0 = NaN for a few exactly-grazing mirror hits (DESIGN.md
    # "Non-finite samples"); such pixels are excluded from the RMSE and counted
    fin = np.isfinite(cpu_film).all(axis=2) & np.isfinite(gpu_film).all(axis=2)
    d = gpu_film[fin].astype(np.float64) - cpu_film[fin].astype(np.float64)
    rmse = float(np.sqrt(np.mean(d * d)))
    return {
        "value": n / dt / 1e6, "unit": "Msamples/s", "cores": threads, "kind": "port",
        "sample": "same scene/integrator, %dx%d at %d spp (%.1f s of CPU work, OpenMP %d threads)" % (params.width, params.height, spp, dt, threads),
    }, {"rmse_gpu_vs_cpu": rmse, "rmse_spp": spp, "rmse_excluded_nonfinite_pixels": int((~fin).sum())}


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

    scene, params, name = workload(args)
    film = torch.zeros((params.height, params.width, 3), dtype=torch.float32, device=dev) if rank == 0 else None

    def barrier():
        if world > 1:
            tdist.barrier()
        torch.cuda.synchronize(dev)

    kernel_ms = []

    def step(record):
        if film is not None:
            film.zero_()
        out = kydist.render_distributed(scene, params, rank, world, local_rank, film=film)
        if record:
            torch.cuda.current_stream(dev).synchronize()
            kernel_ms.append(float(lib.kyhip_kernel_ms(local_rank)))
        return out

    for _ in range(args.warmup):
        step(False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(False)
    barrier()
    elapsed = time.perf_counter() - t0
    # kernel durations: a second, untimed pass with a sync after every step so that each event pair is read back
    for _ in range(max(1, min(args.steps, 3))):
        step(True)
    barrier()

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    k = torch.tensor([sum(kernel_ms) / len(kernel_ms)], dtype=torch.float64, device=dev)
    if world > 1:
        tdist.all_reduce(t, op=tdist.ReduceOp.MAX)
        tdist.all_reduce(k, op=tdist.ReduceOp.MAX)
    elapsed = float(t.item())
    kernel_avg_ms = float(k.item())

    if rank == 0:
        samples_per_step = params.width * params.height * params.samples_per_pixel
        ms_per_step = elapsed / args.steps * 1e3
        value = samples_per_step * args.steps / elapsed / 1e6
        bps = BYTES_PER_SAMPLE[args.workload]
        launch_bytes = bps * samples_per_step / world  # one launch covers this rank's share of the frame
        achieved = launch_bytes / (kernel_avg_ms * 1e-3) / 1e9
        traffic = None
        tp = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tp):
            try:
                with open(tp) as fh:
                    tj = json.load(fh)
                if tj.get("workload") == args.workload and tj.get("samples_per_launch"):
                    traffic = tj["hbm_bytes_per_launch"] * (samples_per_step / world) / tj["samples_per_launch"]
            except Exception:
                traffic = None
        line = {
            "metric": "Msamples/s (paths*spp)", "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": name, "width": params.width, "height": params.height, "spp": params.samples_per_pixel,
                       "max_path_depth": params.max_path_depth, "direct_sample": "both_mis", "seed": params.seed,
                       "tile": [params.tile_w, params.tile_h], "parallelism": "image tiles interleaved over %d GPU(s), one film-tile gather" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "kernel": "render_kernel", "kernel_ms": kernel_avg_ms,
                         "algorithmic_bytes_per_sample": bps, "samples_per_launch": samples_per_step // world},
            "film_mean": float(film.mean().item()),
        }
        if world == 1 and not args.no_cpu_baseline:
            def gpu_film(sample_params):
                return api.render(scene, sample_params, device=local_rank)
            cb, extra = cpu_baseline(scene, params, gpu_film, args.cpu_seconds)
            line["cpu_baseline"] = cb
            line.update(extra)
            line["speedup_vs_cpu_baseline"] = value / cb["value"]
        print(json.dumps(line), flush=True)
    if world > 1:
        tdist.destroy_process_group()


if __name__ == "__main__":
    main()
