#!/bin/bash
# tools/sweep.sh name1 name2 ...  : Cornell (C2) + Veach (C3, 512 spp) bench of each build_variants/<name>.so
for v in "$@"; do
  for wl in "cornell" "veach --spp 512"; do
    KYHIP_LIB=$PWD/build_variants/$v.so python3 bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --no-extra 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('%-16s %-8s %8.1f Msamples/s  kernel %8.2f ms  film_mean %.6f' % ('$v', '$wl'.split()[0], j['value'], j['roofline']['kernel_ms'], j['film_mean']))
"
  done
done
