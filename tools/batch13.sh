#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/b13
python3 -m pytest tests -m gpu -q 2>&1 | tail -3 > gpurun_out/b13/pytest.txt
KYHIP_SHADOW_QUEUE=1 python3 -m pytest tests -m gpu -q 2>&1 | tail -3 >> gpurun_out/b13/pytest.txt
python3 tools/generic_time.py > gpurun_out/b13/generic.txt 2>&1
for wl in cornell veach batch; do python3 bench.py --workload $wl --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('$wl %8.1f Msamples/s  ms/step %8.2f kernel %s' % (j['value'], j['ms_per_step'], ['%.2f' % k for k in j['roofline']['kernel_ms_per_frame']]))
"; done > gpurun_out/b13/bench.txt 2>&1
cat gpurun_out/b13/*.txt
