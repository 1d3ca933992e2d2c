#!/bin/bash
# round 6: the random-room soak (72 rooms) on the table's kernels and on kernels compiled at run time for every room, 
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
{
echo "== KY_RANDOM_ROOMS=72, table kernels (library hash $(python3 -c "from ky_amd import _abi as A; print('%016x' % A.load_kyhip().kyhip_kernel_source_hash())"))"
KY_RANDOM_ROOMS=72 python3 -m pytest tests/test_random_scenes_gpu.py -m gpu -q -s 2>&1 | grep -v amdgpu.ids | tail -90
echo "== KY_RANDOM_ROOMS=72 KYHIP_JIT=1"
KY_RANDOM_ROOMS=72 KYHIP_JIT=1 python3 -m pytest tests/test_random_scenes_gpu.py -m gpu -q -s 2>&1 | grep -v amdgpu.ids | tail -90
} > gpurun_out/r06/soak.txt 2>&1
tail -5 gpurun_out/r06/soak.txt; grep -c . gpurun_out/r06/soak.txt
