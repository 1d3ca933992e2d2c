#!/bin/bash
# tools/sweep_batch.sh name1 name2 ... : the six-frame batch of configs[3] at 256 spp of each build_variants/<name>.so
for v in "$@"; do
  KYHIP_LIB=$PWD/build_variants/$v.so python3 bench.py --workload batch --spp 256 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('%-10s batch %8.1f Msamples/s  kernel %8.2f ms  film_mean %.6f' % ('$v', j['value'], j['roofline']['kernel_ms'], j['film_mean']))
"
done
