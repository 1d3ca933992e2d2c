#!/usr/bin/env python3
"""Per-step host/launch overhead of the device-resident render path: wall time per step vs kernel time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ky_amd import api, dist, _abi as A
lib = A.load_kyhip()
dev = torch.device("cuda", 0)
scene = api.cornell_box_scene(A.CB_DEFAULT_SCENE, 1024, 768)
for spp in (8, 64, 128):
    p = api.make_params(1024, 768, spp)
    film = torch.zeros((768, 1024, 3), device=dev)
    for _ in range(3): dist.render_distributed(scene, p, 0, 1, 0, film=film)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); n = 20
    for _ in range(n):
        film.zero_(); dist.render_distributed(scene, p, 0, 1, 0, film=film)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n * 1e3
    print("spp %4d: %.3f ms/step wall, kernel %.3f ms, overhead %.3f ms" % (spp, dt, lib.kyhip_kernel_ms(0), dt - lib.kyhip_kernel_ms(0)))
