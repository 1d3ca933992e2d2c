#!/bin/bash
# tools/ab6.sh "<bench.py workload args>" name1 name2 ...  : kernel time of each build_variants/<name>.so on one workload, one line each
# (single stream, 3 timed steps + 1 warm-up; the first name is usually repeated at the end to show the box's drift)
ARGS="$1"; shift
for v in "$@"; do
  KYHIP_LIB=$PWD/build_variants/$v.so python3 bench.py $ARGS --steps 3 --warmup 1 --no-pipeline --no-cpu-baseline --no-extra 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('%-14s %-28s %9.1f Msamples/s  kernel %9.3f ms  film_mean %.7f' % ('$v', '$ARGS', j['value'], j['roofline']['kernel_ms'], j['film_mean']))
"
done
