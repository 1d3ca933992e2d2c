#!/bin/bash
# round 5: the box traversal against the per-rectangle one (KYHIP_BOXES=0), same library: tools/r05_box.sh <variant> [pytest -k expression]
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
V=$1; K=${2:-scene_intersect}
{
for B in 1 0 1 0; do
  KYHIP_BOXES=$B KYHIP_LIB=$PWD/build_variants/$V.so python3 bench.py --workload cornell --steps 3 --warmup 1 --no-cpu-baseline --no-extra 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('boxes=$B %8.1f Msamples/s  kernel %8.2f ms  film_mean %.6f  rmse %.3g' % (j['value'], j['roofline']['kernel_ms'], j['film_mean'], j.get('rmse_gpu_vs_cpu') or -1))
"
done
KYHIP_LIB=$PWD/build_variants/$V.so python3 -m pytest tests -m gpu -x -q -k "$K" 2>&1 | tail -8
} > gpurun_out/r05/box_$V.txt 2>&1
cat gpurun_out/r05/box_$V.txt
