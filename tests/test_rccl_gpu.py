"""The RCCL calls of the N > 1 path, executed on hardware with the one world size a single-GPU box allows: one.

RCCL wants a device per rank, so the two-rank tests of this suite (tests/test_configs_gpu.py::test_bench_two_ranks_*) run their collective over gloo, and
`init_process_group("nccl")`, the communicator's barrier / all_gather_object and `torch.distributed.gather` on device tensors -- the calls bench.py and
ky_amd/dist.py make at N > 1 -- stayed unexecuted.  A world of one executes them: the library loads, the communicator is created on the device
bench.py would name (`device_id`), and the gather writes rank 0's own tile buffer into the gather block through the communicator; the film assembled from that
block must be the film `kyhip_render` produces.  What this cannot show is a transfer over xGMI.
"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as tdist
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
tdist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)          # bench.py's call at N > 1
assert tdist.get_backend() == "nccl" and tdist.get_world_size() == 1
ranks = [None]
tdist.all_gather_object(ranks, {"rank": 0, "name": torch.cuda.get_device_properties(dev).name})   # bench.py's `ranks`
assert ranks[0]["rank"] == 0
tdist.barrier()
from ky_amd import api, _abi as A, dist as kd
W, H = 200, 136                                                                   # ragged: 13 x 9 tiles, the last column 8 wide, the last row 8 high
scene = api.cornell_box_scene(A.CB_BOTH_SMALL_SPHERES | A.CB_LIGHT_AREA, W, H)
params = api.make_params(W, H, 32)
fb = kd.frame_buffers(params, 0, 1, dev)
block = torch.full((1,) + tuple(fb.tiles.shape), -1.0, dtype=torch.float32, device=dev)   # a gather block of its own, poisoned
kd.render_shard(scene, params, 0, 1, 0, out=fb.tiles)
got = kd.gather_tiles(fb.tiles, 0, 1, out=block, always_collective=True)          # torch.distributed.gather over RCCL
assert got.data_ptr() == block.data_ptr()
film = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
kd.add_tiles_to_film(film, got, params, 1, 0)
tdist.barrier()
torch.cuda.synchronize(dev)
ref = api.render(scene, params)
f = film.cpu().numpy()
assert np.array_equal(f, ref), float(np.abs(f - ref).max())
assert float(block.min()) >= 0.0                                                   # every slot written by the collective
tdist.destroy_process_group()
print("RCCL_OK", torch.cuda.nccl.version(), float(f.mean()))
'''


def test_rccl_calls_of_the_multi_gpu_path_in_a_world_of_one(tmp_path):
    script = tmp_path / "rccl_one.py"
    script.write_text(SCRIPT % {"root": ROOT})
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29400 + os.getpid() % 300),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, env=env, cwd=tmp_path, timeout=600)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    assert "RCCL_OK" in r.stdout
    print(r.stdout.strip().splitlines()[-1])
