#!/bin/bash
# tools/sweep_generic.sh name1 name2 ... : the run-time-dispatched render kernel (Cornell light_mis, Veach light_mis at 512 spp, Cornell with the
# recursive integrator) of each build_variants/<name>.so
for v in "$@"; do
  for wl in "cornell --direct-sample 32" "veach --spp 512 --direct-sample 32" "cornell --direct-sample 8"; do
    KYHIP_LIB=$PWD/build_variants/$v.so python3 bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('%-10s %-40s %8.1f Msamples/s  kernel %8.2f ms  film_mean %.6f' % ('$v', '$wl', j['value'], j['roofline']['kernel_ms'], j['film_mean']))
"
  done
done
