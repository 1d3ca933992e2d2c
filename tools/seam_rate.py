"""The host-film seam at low spp: kyhip_render (host film in / out) against the device-resident path on configs[1] at 64 spp (bench.py's boundary.rates[1]),
for the thread count of this process (KYHIP_SEAM_THREADS or the library's own choice) under this process's CPU affinity.  tools/seam_cpu_scan.sh sweeps both."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ky_amd import _abi as A, api, dist as kydist
lib = A.load_kyhip()
W, H, spp = 1024, 768, int(sys.argv[1]) if len(sys.argv) > 1 else 64
scene = api.cornell_box_scene(A.CB_DEFAULT_SCENE, W, H); p = api.make_params(W, H, spp)
dev = torch.device("cuda", 0)
film_host = np.zeros((H, W, 3), np.float32); film_dev = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
reps = 30
for _ in range(3): api.render(scene, p, film=film_host)
t0 = time.perf_counter()
for _ in range(reps): api.render(scene, p, film=film_host)
host_ms = (time.perf_counter() - t0) / reps * 1e3
kydist.render_distributed(scene, p, 0, 1, 0, film=film_dev); torch.cuda.synchronize(dev)
t0 = time.perf_counter()
for _ in range(reps):
    kydist.render_distributed(scene, p, 0, 1, 0, film=film_dev); torch.cuda.synchronize(dev)
dev_ms = (time.perf_counter() - t0) / reps * 1e3
print("affinity %3d cpus  seam threads %d  host %.3f ms  device-resident %.3f ms  ratio %.3f  (%s)" % (len(os.sched_getaffinity(0)), lib.kyhip_seam_threads(), host_ms, dev_ms, dev_ms / host_ms, lib.kyhip_multi_status(0).decode()))
