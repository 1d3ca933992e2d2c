"""the headline frame's parity sample at a given tile step: which pixels are off, and what the replay says about them"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("KYHIP_JIT", "0")
import argparse
import numpy as np
import bench
from ky_amd import _abi as A, api
from oracle import kyoracle as O
args = argparse.Namespace(workload="cornell", width=0, height=0, spp=0, depth=0, direct_sample=A.DIRECT_BOTH_MIS, integrator=A.INTEGRATOR_PATH_TRACING_ITERATION)
frames, _, _ = bench.workload(args)
threads = max(1, min(O.max_threads(), bench.cpus_granted()))
for px in [int(a) for a in sys.argv[1:]] or [7936]:
    r = bench.parity_full_spp(frames, lambda scene, sp: api.render(scene, sp, device=0), O, threads, budget_samples=px * 1024)
    print(px, {k: v for k, v in r.items() if k != "review_pixels"})
    for rp in r.get("review_pixels", []):
        print("   review:", rp)
