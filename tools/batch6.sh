#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/b6
python3 -m pytest tests -m gpu -q -x 2>&1 | tail -5 > gpurun_out/b6/pytest_auto.txt
KYHIP_SHADOW_QUEUE=1 python3 -m pytest tests -m gpu -q 2>&1 | tail -8 > gpurun_out/b6/pytest_forced.txt
for q in 0 1; do for wl in cornell "veach --spp 1024"; do
  KYHIP_SHADOW_QUEUE=$q python3 bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('queue=$q %-8s %8.1f Msamples/s  kernel %8.2f ms  film_mean %.6f' % ('$wl'.split()[0], j['value'], j['roofline']['kernel_ms'], j['film_mean']))
"; done; done > gpurun_out/b6/sweep.txt 2>&1
cat gpurun_out/b6/*.txt
